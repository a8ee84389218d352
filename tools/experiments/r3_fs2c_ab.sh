#!/bin/bash
# Three-pass products: column-group size 2^c of the split factor tables (NTTX_FS2_LOGC; 0 = half the slab's log2 width)
for c in 0 5 6 7 8; do
  for k in 24 25 26; do
    echo "== NTTX_FS2_LOGC=$c 2^$k"
    NTTX_FS2_LOGC=$c timeout -k 10 100 python bench.py --workload fourstep --log-len $k --no-cpu --no-context --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1
  done
done
