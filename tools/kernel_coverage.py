"""Diff of the kernels libnttx.so contains against the kernels the traced test runs launched.
usage: python3 tools/kernel_coverage.py <dir holding rocprofv3 --kernel-trace -M outputs>   (markdown on stdout)"""
import collections
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ntt_amd", "csrc", "libnttx.so")


def library_kernels():
    """Mangled names of the kernel descriptors (<name>.kd) in the library's embedded code object."""
    s = subprocess.run(["strings", "-n", "8", LIB], capture_output=True, text=True, check=True).stdout
    return sorted({ln[:-3] for ln in s.splitlines() if ln.startswith("_Z") and ln.endswith(".kd")})


def demangle(names):
    if not names:
        return []
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return [re.sub(r"^void ", "", x) for x in r.stdout.splitlines()]


def family(dem):
    return re.sub(r"[<(].*", "", dem)


def main(out):
    seen = collections.defaultdict(set)        # mangled name -> runs that launched it
    for f in glob.glob(os.path.join(out, "*", "**", "*kernel_trace.csv"), recursive=True):
        run = os.path.relpath(f, out).split(os.sep)[0]
        for r in csv.DictReader(open(f)):
            seen[r["Kernel_Name"].removesuffix(".kd")].add(run)
    lib = library_kernels()
    dem = dict(zip(lib, demangle(lib)))
    missing = [k for k in lib if k not in seen]
    only_children = [k for k in lib if k in seen and "main" not in seen[k]]
    fam_all, fam_miss = collections.Counter(family(dem[k]) for k in lib), collections.Counter(family(dem[k]) for k in missing)
    print("# Kernel coverage of the GPU parity suite\n")
    print(f"`libnttx.so` holds {len(lib)} kernels; {len(lib) - len(missing)} were launched by a run that compares with the oracle "
          f"({len(only_children)} of them only by a measurement-switch child or a harness binary), {len(missing)} by none.\n")
    print("| kernel family | in the library | never launched |\n|---|---|---|")
    for fam in sorted(fam_all):
        print(f"| `{fam}` | {fam_all[fam]} | {fam_miss.get(fam, 0)} |")
    print("\n## Never launched\n")
    for k in missing:
        print(f"- `{dem[k]}`")
    print("\n## Launched only under a measurement switch or by a harness binary\n")
    for k in only_children:
        print(f"- `{dem[k]}` ({', '.join(sorted(seen[k]))})")
    foreign = sorted(k for k in seen if k not in dem and "nttx" in k)
    if foreign:
        print("\n## Traced nttx kernels that are not in the library (stale build?)\n")
        for k in foreign:
            print(f"- `{k}`")


if __name__ == "__main__":
    main(sys.argv[1])
