#!/bin/bash
# per-kernel HIP-event times of the lengths served by the long-row kernels: two-pass 2^21..2^23 and single-pass 2^12..2^14
set -o pipefail
for k in 21 22 23; do
  timeout -k 10 100 python bench.py --workload fourstep --log-len $k --no-cpu --steps 50 --warmup 10 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2^$k', round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1
done
for k in 12 13 14; do
  timeout -k 10 100 python tools/bench_dist.py --workload batch --log-len $k --batch $((1<<(26-k))) --steps 20 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2^$k', round(d['us_per_mult_per_gpu'],4), 'us each', d['kernels_ms_per_step'], d['passes'])" || exit 1
done
