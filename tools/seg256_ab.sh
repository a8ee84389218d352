#!/bin/bash
# A/B of 64-column (256-byte segment) tiles for the 2^6 .. 2^8-point column passes: 2^26 points per step, lengths 2^15 .. 2^17
set -o pipefail
for k in 15 16 17; do
  for s in 0 1; do
    NTTX_COL_SEG256=$s timeout -k 10 100 python tools/bench_dist.py --workload batch --log-len $k --batch $((1<<(26-k))) --steps 20 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2^$k seg256=$s', round(d['ms_per_step'],4), 'ms/step', {k: round(v,4) for k,v in d['kernels_ms_per_step'].items()}, d['passes'])" || exit 1
  done
done
