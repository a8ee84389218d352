#!/bin/bash
# Job wrapper around the harness, the local counterpart of the reference's PBS job (qsub.sh:6-14: copy the program
# and files/ to the node, run it there, copy files/ back).  No scheduler here: the "node" is a scratch directory.
#   tools/launch_harness.sh <data dir with N.in / N.out> <dir that receives files/N.out> [first id] [last id] [int]
# A fifth argument "int" runs the (int, int) flavour (monttry2.cpp's harness shape).
set -euo pipefail
data=${1:?data directory}; dest=${2:?result directory}; first=${3:-0}; last=${4:-4}; flavour=${5:-u64}
here=$(cd "$(dirname "$0")/.." && pwd)
exe=$here/ntt_amd/csrc/nttx_harness; [ "$flavour" = int ] && exe=${exe}_int
work=$(mktemp -d /tmp/nttx_job.XXXXXX)
trap 'rm -rf "$work"' EXIT
mkdir -p "$work/files" "$dest"
start=$(date +%s.%N)
NTT_DATA_DIR=$data NTT_OUT_DIR=$work/files "$exe" "$first" "$last" | tee "$work/test.o"
rc=${PIPESTATUS[0]}
cp "$work"/files/*.out "$dest"/ 2>/dev/null || true
cp "$work/test.o" "$dest/test.o"
echo "job finished in $(echo "$(date +%s.%N) - $start" | bc) s, exit code $rc" | tee -a "$dest/test.o"
exit "$rc"
