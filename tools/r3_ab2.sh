#!/bin/bash
# CRT route (n = 2^17, four primes): residue split fused into the first column pass (default) against a split pass of its own
out=gpurun_out/r3c; mkdir -p $out; rm -f $out/crt_ab.txt
for B in 1 8 32 128; do
  for f in 1 0; do
    echo "B=$B fused=$f" >> $out/crt_ab.txt
    NTTX_CRT_FUSED_SPLIT=$f B=$B K=40 timeout -k 10 120 python tools/bench_crt.py 2>/dev/null | head -1 >> $out/crt_ab.txt
  done
done
cat $out/crt_ab.txt
