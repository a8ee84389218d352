#!/bin/bash
# Three-pass products: intermediate vectors with padded outer-pass rows (default) against dense ones (NTTX_PAD3=0)
for v in "NTTX_PAD3=0" "NTTX_PAD3=1"; do
  for k in 24 25 26; do
    echo "== $v 2^$k"
    env $v timeout -k 10 100 python bench.py --workload fourstep --log-len $k --no-cpu --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1
  done
done
