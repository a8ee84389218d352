#!/bin/bash
# A/B of the arithmetic classes for small primes: NTTX_WIDE=0 (signed rows with 4 multiples of headroom) against the default
# (16 multiples), products of length 2^18 / 2^16 / 2^13, batch = 2^26 points; an optional experiment library as third leg
set -o pipefail
for p in 104857601 7340033; do
  for w in 0 1 $1; do
    lib=""; ww=$w; if [ "$w" != 0 ] && [ "$w" != 1 ]; then lib=$w; ww=1; fi
    for k in 18 16 13; do
      NTTX_LIB=$lib NTTX_WIDE=$ww timeout -k 10 100 python tools/bench_dist.py --workload batch --p $p --log-len $k --batch $((1<<(26-k))) --steps 20 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('p=$p wide=$w 2^$k', round(d['ms_per_step'],4), 'ms/step', {k: round(v,4) for k,v in d['kernels_ms_per_step'].items()}, d['passes'])" || exit 1
    done
  done
done
