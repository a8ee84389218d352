#!/bin/bash
# 64-bit-lane column passes: workgroup shapes against each other (A/B through the measurement switches of nttx64_col.hip).
# Batch 128 products of 2^18 points, modulus 19 * 2^46 + 1 (floating-point class).
out=gpurun_out/r3d; mkdir -p $out; rm -f $out/col64_ab.txt
run() { echo "$1" >> $out/col64_ab.txt; env $1 timeout -k 10 100 python tools/bench_direct64.py $2 2>/dev/null >> $out/col64_ab.txt || exit 1; }
for v in "$@"; do run "$v" "18:128"; done
cat $out/col64_ab.txt
