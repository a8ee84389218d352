#!/bin/bash
# per-kernel times of a 2^18 x 256 product batch for one prime of every arithmetic class (and the four CRT primes)
set -o pipefail
for p in 7340033 469762049 998244353 1004535809 1224736769; do
  timeout -k 10 100 python tools/bench_dist.py --workload batch --p $p --log-len 18 --batch 256 --steps 20 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('p=$p', round(d['ms_per_step'],4), 'ms/step', {k: round(v,4) for k,v in d['kernels_ms_per_step'].items()}, d['passes'])" || exit 1
done
