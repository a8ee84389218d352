#!/bin/bash
# Three-pass products, same box, two rounds: the library of the previous commit against the current one with dense
# (NTTX_PAD3=0) and padded intermediate vectors
run() { env NTTX_LIB=$PWD/$1 $2 timeout -k 10 100 python bench.py --workload fourstep --log-len $3 --no-cpu --no-context --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1; }
for rep in 1 2; do
  for k in 24 25 26; do
    echo "== head 2^$k";          run tools/ablate/libnttx_head.so NTTX_X=0 $k
    echo "== new, dense 2^$k";    run ntt_amd/csrc/libnttx.so NTTX_PAD3=0 $k
    echo "== new, padded 2^$k";   run ntt_amd/csrc/libnttx.so NTTX_PAD3=1 $k
  done
done
