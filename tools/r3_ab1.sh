#!/bin/bash
# Round-3 measurements, first batch: launch / synchronisation costs, host-memory access forms, the CRT route with the residue
# split fused into the first column pass against the split as a pass of its own, the host-buffer call.
out=gpurun_out/r3c; mkdir -p $out
timeout -k 10 120 tools/ubench_sync > $out/sync.txt 2>&1 || { echo sync failed; tail $out/sync.txt; exit 1; }
timeout -k 10 120 tools/ubench_hostmem > $out/hostmem.txt 2>&1 || { echo hostmem failed; tail $out/hostmem.txt; exit 1; }
for B in 1 8 32 128; do
  for f in 1 0; do
    echo "B=$B fused=$f" >> $out/crt_ab.txt
    NTTX_CRT_FUSED_SPLIT=$f B=$B K=40 timeout -k 10 120 python tools/bench_crt.py 2>&1 | head -1 >> $out/crt_ab.txt || exit 1
  done
done
g++ -O2 -std=c++17 tools/host_latency.cpp -Iinclude -Lntt_amd/csrc -lnttx -Wl,-rpath,$PWD/ntt_amd/csrc -o tools/host_latency_bin && timeout -k 10 120 tools/host_latency_bin > $out/host_latency.txt 2>&1
cat $out/sync.txt $out/hostmem.txt $out/crt_ab.txt $out/host_latency.txt
