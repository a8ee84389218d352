#!/bin/bash
# Which kernels of libnttx.so does an oracle comparison launch?  (run on the GPU box from the repository root)
#   tools/kernel_coverage.sh [outdir]      -> <outdir>/coverage.md (copy it to profiles/ to keep it)
# The in-process GPU tests are traced in one run; every test that starts a child program is left out of that run and its
# child traced in a run of its own, with the child itself as the program after "--" (the profiler initialises the GPU
# before the traced program starts, so nothing may exec under it).
set -u
out=${1:-gpurun_out/cov}
rm -rf "$out"; mkdir -p "$out"
root=$(pwd)
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd "$root"
TRACE="rocprofv3 --kernel-trace -M --output-format csv"

echo "== in-process tests" | tee "$out/log.txt"
timeout -k 10 1100 $TRACE -d "$out/main" -- python3 -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread \
  -k "not child_process and not harness_binary and not three_pass_plan_at_2_22 and not rccl and not bench_two_ranks" \
  > "$out/main.log" 2>&1 || { echo "in-process tests failed"; tail -30 "$out/main.log"; exit 1; }
tail -2 "$out/main.log" | tee -a "$out/log.txt"

# children of tests/test_gpu_dispatch.py: one traced run per measurement switch
python3 - "$out" <<'PY' > "$out/children.txt"
import sys
sys.path.insert(0, ".")
from tests.test_gpu_dispatch import SWITCHES
for i, (env, suites) in enumerate(SWITCHES):
    print(i, " ".join(f"{k}={v}" for k, v in env.items()), "--", " ".join(suites))
PY
while read -r idx rest; do
  envs=${rest%% -- *}; suites=${rest##* -- }
  echo "== child $idx: $envs : $suites" | tee -a "$out/log.txt"
  ( export $envs; timeout -k 10 280 $TRACE -d "$out/child_$idx" -- python3 -u tests/sweeps.py $suites > "$out/child_$idx.log" 2>&1 ) \
    || { echo "child $idx failed"; tail -20 "$out/child_$idx.log"; exit 1; }
done < "$out/children.txt"

# the two harness binaries on ids 0..4 (data written by the oracle first, untraced)
python3 tools/make_harness_data.py "$out/nttdata" > "$out/harness_data.log" 2>&1 || { tail "$out/harness_data.log"; exit 1; }
mkdir -p "$out/files"
export NTT_DATA_DIR="$root/$out/nttdata" NTT_OUT_DIR="$root/$out/files"
echo "== harness binaries" | tee -a "$out/log.txt"
timeout -k 10 200 $TRACE -d "$out/harness" -- ntt_amd/csrc/nttx_harness > "$out/harness.log" 2>&1 || { tail "$out/harness.log"; exit 1; }
timeout -k 10 200 $TRACE -d "$out/harness_int" -- ntt_amd/csrc/nttx_harness_int 0 3 > "$out/harness_int.log" 2>&1 || { tail "$out/harness_int.log"; exit 1; }
grep -c "结果正确" "$out/harness.log" "$out/harness_int.log" | tee -a "$out/log.txt"

python3 tools/kernel_coverage.py "$out" > "$out/coverage.md" || exit 1
# the raw traces are large: keep the summary and the logs only
rm -rf "$out/main" "$out"/child_[0-9] "$out"/child_[0-9][0-9] "$out/harness" "$out/harness_int" "$out/nttdata" "$out/files"
head -40 "$out/coverage.md"
