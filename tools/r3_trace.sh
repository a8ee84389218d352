#!/bin/bash
# kernel trace of one bench.py workload: per-kernel average durations (which pass of a three-pass plan costs what)
# usage: tools/r3_trace.sh <name> <bench.py arguments>
name=$1; shift
out=gpurun_out/trace_$name; rm -rf $out; mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 bench.py "$@" --no-cpu --no-context > $out/log.txt 2>&1 || { tail $out/log.txt; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/t/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "nttx::" in r["Name"]:
            print(r["Name"].split("(")[0].replace("void nttx::", ""), r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1))
PY
