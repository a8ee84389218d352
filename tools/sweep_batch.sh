#!/bin/bash
# tools/sweep_batch.sh KNOB v1 v2 ... : the 4096 x 2^16 / 2^17 batch workloads under an environment knob
knob=$1; shift
for g in "$@"; do
  for ll in 16 17; do
    env $knob=$g timeout -k 10 200 python tools/bench_dist.py --workload batch --log-len $ll --steps 20 --warmup 5 > gpurun_out/_s.log 2>/dev/null || exit 1
    python - "$knob" "$g" <<'PY' >> gpurun_out/sweep_batch.log
import json, sys
d = json.loads(open("gpurun_out/_s.log").readline())
print(sys.argv[1], sys.argv[2], d["log2_len"], d["passes"], round(d["us_per_mult_per_gpu"], 4), round(d["model_GBs_per_gpu"]))
PY
  done
done
