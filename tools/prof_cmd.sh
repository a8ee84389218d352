#!/bin/bash
# counter passes over any benchmark command of this repository: tools/prof_cmd.sh <tag> <python script and arguments>
tag=$1; shift
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 "$@" > $out/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq -- python3 "$@" > $out/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq2 -- python3 "$@" > $out/pmc_sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 "$@" > $out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 "$@" > $out/pmc_write.log 2>&1 || exit 1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "nttx::" in r["Kernel_Name"]:
                d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        print(k, {c: round(sum(x) / len(x) / 1e6, 3) for c, x in v.items()}, "(millions per launch)")
for f in glob.glob(f"{out}/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "nttx::" in r["Name"]:
            print(r["Name"].split("(")[0], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3, "min us", float(r["MinNs"]) / 1e3)
PY
