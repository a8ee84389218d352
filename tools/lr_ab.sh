#!/bin/bash
# A/B of long-row kernel builds: the in-tree library against NTTX_LIB=<experiment build>
# usage: tools/lr_ab.sh <experiment .so> "<single-pass log lengths>" "<two-pass log lengths>"
set -o pipefail
for lib in "" "$1"; do
  echo "== lib: ${lib:-default}"
  for k in $2; do
    NTTX_LIB=$lib timeout -k 10 100 python tools/bench_dist.py --workload batch --log-len $k --batch $((1<<(26-k))) --steps 20 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2^$k', round(d['us_per_mult_per_gpu'],4), 'us each', d['kernels_ms_per_step'], d['passes'])" || exit 1
  done
  for k in $3; do
    NTTX_LIB=$lib timeout -k 10 100 python bench.py --workload fourstep --log-len $k --no-cpu --steps 50 --warmup 10 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2^$k', round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1
  done
done
