#!/bin/bash
# usage: tools/sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each arg = one bench configuration)
for cfg in "$@"; do
  env $cfg timeout -k 10 100 python bench.py --steps 20 --no-cpu $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print('$cfg', '| us/mult', round(d['us_per_mult_per_gpu'],3), '| ms/step', round(d['ms_per_step'],4), {a:round(b,4) for a,b in k.items()})"
done
