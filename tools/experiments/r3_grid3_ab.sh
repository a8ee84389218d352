#!/bin/bash
# Three-pass products: workgroups per column-pass launch (one tile each by default; fewer = resident workgroups striding
# over the tiles with the next tile's loads in flight during the butterflies)
for v in "NTTX_GRID_COL=1048576" "NTTX_GRID_COL=4096 NTTX_GRID_COLI=2048" "NTTX_GRID_COL=2048 NTTX_GRID_COLI=1024" "NTTX_GRID_COL=1024 NTTX_GRID_COLI=512"; do
  for k in 24 26; do
    echo "== $v 2^$k"
    env $v timeout -k 10 100 python bench.py --workload fourstep --log-len $k --no-cpu --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1
  done
done
