#!/bin/bash
# one bench.py line per BASELINE configuration at one GPU -> gpurun_out/r2rec (copied to profiles/ afterwards)
set -o pipefail
out=gpurun_out/r2rec; mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $out/$name.json 2> $out/$name.err || { tail $out/$name.err; exit 1; }; }
run c2_single
run c3_crt --workload crt
run c3_crt_batch128 --workload crt --batch 128 --no-cpu
run c4_batch_2p16 --workload batch
run c4_batch_2p17 --workload batch --log-len 17
run c5_fourstep_2p24 --workload fourstep
run c5_fourstep_2p25 --workload fourstep --log-len 25 --no-cpu
run c5_fourstep_2p26 --workload fourstep --log-len 26 --no-cpu
for f in $out/*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], "value", round(d["value"], 2), d["unit"], "| ms/step", round(d["ms_per_step"], 4), "| dom", str(r.get("kernel"))[:20], "frac", round(r.get("frac", 0), 3),
      "whole", round((r.get("whole_step") or {}).get("frac", 0), 3), "| cpu", round((d.get("cpu_baseline") or {}).get("value", 0), 3))
PY
done
