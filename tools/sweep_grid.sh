#!/bin/bash
# sweep an environment knob over the CRT and headline benches: tools/sweep_grid.sh KNOB v1 v2 ...
knob=$1; shift
for g in "$@"; do
  env $knob=$g B=128 timeout -k 10 200 python tools/bench_crt.py > gpurun_out/_s.log 2>/dev/null || exit 1
  python - "$knob" "$g" <<'PY' >> gpurun_out/sweep_grid.log
import json, sys
d = json.loads(open("gpurun_out/_s.log").readline())
print("crt", sys.argv[1], sys.argv[2], round(d["us_per_mult"], 3), d["kernels_ms_per_step"])
d = json.loads(open("gpurun_out/_s.log").readlines()[1])
print("direct64", sys.argv[1], sys.argv[2], round(d["us_per_mult"], 3), d["kernels_ms_per_step"])
PY
  env $knob=$g timeout -k 10 200 python bench.py --no-cpu > gpurun_out/_s.log 2>/dev/null || exit 1
  python - "$knob" "$g" <<'PY' >> gpurun_out/sweep_grid.log
import json, sys
d = json.loads(open("gpurun_out/_s.log").readline())
print("headline", sys.argv[1], sys.argv[2], round(d["us_per_mult_per_gpu"], 3), {k: round(v, 4) for k, v in d["kernels_ms_per_step"].items()})
PY
done
