#!/usr/bin/env python3
"""Turn a tools/prof.sh result directory (gpurun_out/prof_<tag>) into the committed summary profiles/<tag>_rocprof.md and
merge its per-kind counter figures into profiles/traffic.json and profiles/valu.json under the workload's key.

  python3 tools/summarize_prof.py <tag> [<counter_record_key>]      (key default: read from the traced bench line)

HBM bytes follow the MI355X guide's recipe: FETCH_SIZE and WRITE_SIZE come from separate --pmc passes, are reported in
KiB, and FETCH_SIZE counts a 128-byte request of a wide (16 B per lane) coalesced stream as 64 bytes, so it is doubled
for these kernels (their loads are all 16 B per lane; the doubled value matches the algorithmic byte count to 1 %).
Kernels are grouped into the kinds bench.py times (col_fwd, row_mul, col_inv, split, crt); a kind's figure is the mean
over all launches of that kind in the run."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join("gpurun_out", f"prof_{tag}")
os.makedirs("profiles", exist_ok=True)
line = {}
try:
    line = json.loads(open(os.path.join(src, "bench_line.json")).read())
except Exception:                                          # noqa: BLE001
    pass
key = sys.argv[2] if len(sys.argv) > 2 else line.get("counter_record_key")
if not key:
    sys.exit("no counter_record_key: pass it as the second argument")
command = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "?"


def short(name):
    return name.split("(")[0].replace("void ", "").replace("nttx::", "")


def kind_of(name):
    m = re.match(r"col_kernel<\d+, (true|false)", name)
    if m:
        return "col_inv" if m.group(1) == "true" else "col_fwd"
    if name.startswith(("row_mul9_kernel", "longrow_mul_kernel")) or re.match(r"row_kernel<\d+, 2", name):
        return "row_mul"
    if name.startswith("split_kernel"):
        return "split"
    if name.startswith("crt_kernel"):
        return "crt"
    return None


def pmc(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "nttx::" in r["Kernel_Name"]:
                d[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


stats = []
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "nttx::" in r["Name"]:
            stats.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"]), float(r["Percentage"])))

fetch, write, sq, sq2 = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq"), pmc("pmc_sq2")
mean = lambda v: sum(v) / len(v) if v else float("nan")                                   # noqa: E731
lines = [f"# rocprofv3 summary `{tag}` — `{command}` (counter record key `{key}`)", ""]
if line:
    lines += [f"Traced run's own line: {line.get('value', 0):.1f} {line.get('unit', '')}, {line.get('ms_per_step', 0):.4f} ms per step "
              "(under the profiler: slower than an unprofiled run).", ""]
lines += ["Kernel trace (`rocprofv3 --kernel-trace --stats`):", "",
          "| kernel | kind | calls | avg us | min us | max us | % of GPU time |", "|---|---|---|---|---|---|---|"]
for n, c, a, mn, mx, pct in stats:
    lines.append(f"| `{n}` | {kind_of(n) or '-'} | {c} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {pct:.1f} |")
lines += ["", "Counters (one `--pmc` pass each; values are per launch, mean over launches):", "",
          "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes (2*FETCH+WRITE) | SQ_INSTS_VALU | SQ_INSTS_LDS | SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE | SQ_WAVE_CYCLES | SQ_WAIT_ANY | SQ_WAIT_INST_ANY | GRBM_GUI_ACTIVE/8 |",
          "|---|---|---|---|---|---|---|---|---|---|---|"]
# per kind: sums over the run of (value x launches), divided by the kind's launches
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for n, c, a, *_ in stats:
    f, w = mean(fetch[n]["FETCH_SIZE"]), mean(write[n]["WRITE_SIZE"])
    hbm = (2 * f + w) * 1024
    s, s2 = sq[n], sq2[n]
    k = kind_of(n)
    if k:
        A = acc[k]
        A["calls"] += c
        A["ns"] += c * a
        for name_, v in (("hbm", hbm), ("valu", mean(s["SQ_INSTS_VALU"])), ("gui", mean(s2["GRBM_GUI_ACTIVE"]))):
            if v == v:
                A[name_] += c * v
                A[name_ + "_calls"] += c
    lines.append(f"| `{n}` | {f:.0f} | {w:.0f} | {hbm/1e6:.1f} MB | {mean(s['SQ_INSTS_VALU'])/1e6:.2f} M | {mean(s['SQ_INSTS_LDS'])/1e6:.2f} M | "
                 f"{mean(s['SQ_LDS_BANK_CONFLICT'])/1e6:.2f} M / {mean(s['SQ_LDS_IDX_ACTIVE'])/1e6:.2f} M | {mean(s['SQ_WAVE_CYCLES'])/1e6:.1f} M | "
                 f"{mean(s2['SQ_WAIT_ANY'])/1e6:.1f} M | {mean(s2['SQ_WAIT_INST_ANY'])/1e6:.1f} M | {mean(s2['GRBM_GUI_ACTIVE'])/8/1e3:.1f} k cycles |")
traffic, valu_insts, clocks, avg_us = {}, {}, {}, {}
lines += ["", "Per kind (mean per launch; 1024 SIMDs; the VALU share is a MODEL: instructions x 4.6 cycles, the issue cost of butterfly "
          "code measured alone; GRBM_GUI_ACTIVE / 8 over the traced duration reads high on launches shorter than about 0.3 ms):", "",
          "| kind | launches | avg us | HBM bytes | TB/s | VALU instructions | clock GHz | VALU issue share by the model |", "|---|---|---|---|---|---|---|---|"]
for k, A in acc.items():
    a_ns = A["ns"] / A["calls"]
    avg_us[k] = a_ns / 1e3
    if A["hbm_calls"]:
        traffic[k] = A["hbm"] / A["hbm_calls"]
    if A["valu_calls"]:
        valu_insts[k] = A["valu"] / A["valu_calls"]
    if A["gui_calls"]:
        clocks[k] = A["gui"] / A["gui_calls"] / 8 / (a_ns * 1e-9) / 1e9
    share = valu_insts.get(k, float("nan")) * 4.6 / 1024 / (clocks.get(k, float("nan")) * 1e9 * a_ns * 1e-9)
    lines.append(f"| {k} | {int(A['calls'])} | {a_ns/1e3:.1f} | {traffic.get(k, float('nan'))/1e6:.1f} MB | {traffic.get(k, float('nan'))/(a_ns*1e-9)/1e12:.2f} | "
                 f"{valu_insts.get(k, float('nan'))/1e6:.2f} M | {clocks.get(k, float('nan')):.2f} | {share*100:.0f} % |")
open(os.path.join("profiles", f"{tag}_rocprof.md"), "w").write("\n".join(lines) + "\n")

# the records bench.py quotes: tagged with a hash of the kernel sources, refused there once the kernels change
sys.path.insert(0, os.getcwd())
from bench import kernel_source_tag  # noqa: E402
now = kernel_source_tag()
for fname, what, payload in (
        ("traffic.json", "HBM bytes per launch and kind: (2 * FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes, mean over the kind's launches",
         {"kernels": traffic}),
        ("valu.json", "SQ_INSTS_VALU per launch and kind (wave-level vector instructions); clock_GHz = GRBM_GUI_ACTIVE / 8 / kernel duration "
                      "in the counter and trace passes", {"kernels": valu_insts, "clock_GHz": clocks})):
    path = os.path.join("profiles", fname)
    try:
        rec = json.load(open(path))
    except Exception:                                      # noqa: BLE001
        rec = {}
    if rec.get("kernel_source_tag") != now or "workloads" not in rec:
        rec = {"kernel_source_tag": now, "what": what, "workloads": {}}        # records of other kernel sources are dropped
    rec["workloads"][key] = dict(payload, measured_by=f"tools/prof.sh {tag}: {command}", avg_us_in_trace=avg_us)
    json.dump(rec, open(path, "w"), indent=1)
print("\n".join(lines))
