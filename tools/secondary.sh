#!/bin/bash
# Secondary workloads on one GPU, one JSON line each: tools/secondary.sh <outfile>
out=${1:-gpurun_out/secondary.jsonl}
: > "$out"
run() { timeout -k 10 150 python tools/bench_dist.py "$@" 2>/dev/null | grep '^{' >> "$out" || exit 1; }
run --workload crt --batch 32 --steps 30 --warmup 20
run --workload crt --batch 128 --steps 20 --warmup 10
for cfg in "12 16384" "13 8192" "14 4096" "16 4096" "17 4096" "20 256" "21 64" "22 32" "23 16"; do
  set -- $cfg
  run --workload batch --log-len $1 --batch $2 --steps 30 --warmup 20
done
for k in 24 25 26; do run --workload fourstep --log-len $k --steps 20 --warmup 10; done
