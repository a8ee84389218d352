#!/bin/bash
# One product of 2^21 .. 2^23 points: a 2^9-point column pass around long rows (default) against three passes (NTTX_LONGROW=2)
for v in 3 2; do
  for k in 21 22 23; do
    echo "== NTTX_LONGROW=$v 2^$k"
    NTTX_LONGROW=$v timeout -k 10 100 python bench.py --workload fourstep --log-len $k --no-cpu --no-context --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1
  done
done
