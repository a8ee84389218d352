#!/bin/bash
# Round-3 records on the GPU box: counter passes per workload (tagged traffic / VALU records that bench.py then quotes),
# one bench.py line per BASELINE configuration, batch-1 and host-buffer latencies.
#   tools/r3_records.sh prof    -> gpurun_out/prof_r3_<workload>/ + profiles/r3_rocprof_<workload>.md, profiles/traffic.json, valu.json
#   tools/r3_records.sh bench   -> gpurun_out/r3rec/*.json (copy to profiles/r3_bench/)
set -o pipefail
mode=${1:-bench}
if [ "$mode" = prof ]; then
  prof() { name=$1; shift; tools/prof.sh r3_$name "$@" > gpurun_out/prof_r3_$name.log 2>&1 || { tail gpurun_out/prof_r3_$name.log; exit 1; }
           python3 tools/summarize_prof.py r3_$name > /dev/null || exit 1; echo "prof $name done"; }
  prof single --steps 200 --warmup 50
  prof crt --workload crt --steps 20 --warmup 5
  prof crt_b128 --workload crt --batch 128 --steps 10 --warmup 3
  prof batch --workload batch --steps 20 --warmup 5
  prof fourstep --workload fourstep --steps 20 --warmup 5
  prof fourstep26 --workload fourstep --log-len 26 --steps 20 --warmup 5
  cp profiles/traffic.json profiles/valu.json gpurun_out/
  for n in single crt crt_b128 batch fourstep fourstep26; do cp profiles/r3_$n\_rocprof.md gpurun_out/; done
  exit 0
fi
out=gpurun_out/r3rec; mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $out/$name.json 2> $out/$name.err || { tail $out/$name.err; exit 1; }; }
run c2_single
run c3_crt --workload crt
run c3_crt_batch128 --workload crt --batch 128 --no-cpu
run c4_batch_2p16 --workload batch
run c4_batch_2p17 --workload batch --log-len 17 --no-cpu
run c5_fourstep_2p24 --workload fourstep
run c5_fourstep_2p25 --workload fourstep --log-len 25 --no-cpu
run c5_fourstep_2p26 --workload fourstep --log-len 26 --no-cpu
for f in $out/*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], "value", round(d["value"], 2), d["unit"], "| ms/step", round(d["ms_per_step"], 4), "| bound", r.get("bound"), "dom", str(r.get("kernel"))[:20], "frac", round(r.get("frac", 0), 3),
      "whole", round((r.get("whole_step") or {}).get("frac", 0), 3), "traffic", r.get("traffic"), "| cpu", round((d.get("cpu_baseline") or {}).get("value", 0), 3))
PY
done
for k in 15 16 17 18 19 20; do timeout -k 10 60 python tools/lat1.py $k 300 2>/dev/null; done > $out/batch1_latency.txt
g++ -O2 -std=c++17 tools/host_latency.cpp -Iinclude -Lntt_amd/csrc -lnttx -Wl,-rpath,$PWD/ntt_amd/csrc -o tools/host_latency_bin && timeout -k 10 120 tools/host_latency_bin > $out/host_latency.txt 2>&1
cat $out/batch1_latency.txt $out/host_latency.txt
