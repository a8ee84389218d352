#!/bin/bash
# What the outer (split-table) column pass of a three-pass product waits for: measurement builds of nttx_col.hip with
# -DNTTX_ABLATE_FS2=<bits> (1 no factor multiplications, 2 no butterflies, 4 no loads, 8 no stores, 16 high-part factor
# from a register instead of its table, 32 low-part factors likewise; results are wrong by design), libraries under
# tools/ablate/.  The inner pass and the row pass are the shipped kernels in every build.
for n in ${ABL_LIST:-0 1 2 3 4 8 11}; do
  lib=tools/ablate/libnttx_abl$n.so; [ $n = 0 ] && lib=ntt_amd/csrc/libnttx.so
  for k in 24 26; do
    echo "== ablate $n 2^$k"
    NTTX_LIB=$PWD/$lib timeout -k 10 100 python bench.py --workload fourstep --log-len $k --no-cpu --no-check --no-context --steps 30 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1
  done
done
