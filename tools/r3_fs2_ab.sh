#!/bin/bash
# Three-pass products: size of the outer pass (NTTX_PLAN3_OUTER, measurement build)
for v in "NTTX_PLAN3_OUTER=0" "NTTX_PLAN3_OUTER=7" "NTTX_PLAN3_OUTER=8" "NTTX_PLAN3_OUTER=9"; do
  for k in 24 25 26; do
    echo "== $v 2^$k"
    env $v timeout -k 10 100 python bench.py --workload fourstep --log-len $k --no-cpu --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1
  done
done
