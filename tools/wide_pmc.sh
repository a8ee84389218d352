#!/bin/bash
# VALU instruction counts per launch of the three kernels of a 2^18 product (batch 256) in both arithmetic-class settings
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
for w in 0 1; do
  out=gpurun_out/wide_pmc_$w; rm -rf $out; mkdir -p $out
  NTTX_WIDE=$w timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out -- python3 tools/bench_dist.py --workload batch --p 104857601 --log-len 18 --batch 256 --steps 10 > $out/log.txt 2>&1 || exit 1
  python3 - $out $w <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "nttx::" in r["Kernel_Name"] and ("col_kernel" in r["Kernel_Name"] or "row_mul" in r["Kernel_Name"]):
            d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print("wide=" + sys.argv[2], k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()})
PY
done
