#!/bin/bash
# record run on the GPU box: rocprofv3 passes over the headline bench, then one bench line per workload
set -o pipefail
out=gpurun_out/r2final; mkdir -p $out
bash tools/prof.sh r2 > $out/prof.log 2>&1 || { tail -20 $out/prof.log; exit 1; }
python tools/summarize_prof.py r2 > $out/summary.txt 2>&1 || { tail -20 $out/summary.txt; exit 1; }
cp profiles/r2_rocprof.md profiles/traffic.json profiles/valu.json $out/
timeout -k 10 300 python bench.py > $out/bench_single.json 2> $out/bench_single.err || { tail $out/bench_single.err; exit 1; }
timeout -k 10 300 python bench.py --workload crt > $out/bench_crt.json 2> $out/err || { tail $out/err; exit 1; }
timeout -k 10 300 python bench.py --workload crt --batch 128 --no-cpu > $out/bench_crt_b128.json 2> $out/err || { tail $out/err; exit 1; }
timeout -k 10 300 python bench.py --workload batch > $out/bench_batch16.json 2> $out/err || { tail $out/err; exit 1; }
timeout -k 10 300 python bench.py --workload batch --log-len 17 --no-cpu > $out/bench_batch17.json 2> $out/err || { tail $out/err; exit 1; }
timeout -k 10 300 python bench.py --workload fourstep > $out/bench_fourstep24.json 2> $out/err || { tail $out/err; exit 1; }
for k in 25 26; do timeout -k 10 300 python bench.py --workload fourstep --log-len $k --no-cpu > $out/bench_fourstep$k.json 2> $out/err || { tail $out/err; exit 1; }; done
for k in 15 16 17 18 19 20; do timeout -k 10 60 python tools/lat1.py $k 300 2>/dev/null | grep log2 >> $out/lat1.txt; done
timeout -k 10 100 tools/host_latency_bin > $out/host_latency.txt 2>&1
cat $out/summary.txt | tail -12; cat $out/lat1.txt $out/host_latency.txt
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], "value", round(d["value"], 2), d["unit"], "| ms/step", round(d["ms_per_step"], 4), "| dom", r.get("kernel"), "frac", round(r.get("frac", 0), 3),
      "whole", round((r.get("whole_step") or {}).get("frac", 0), 3), "| cpu", round((d.get("cpu_baseline") or {}).get("value", 0), 3))
PY
done
