#!/bin/bash
# A/B of the long-row plans: NTTX_LONGROW=0 (plans without them) against 3 (default) on one GPU.
# usage: tools/sweep_longrow.sh <outfile>
out=${1:-gpurun_out/longrow.jsonl}
: > "$out"
for mode in 0 3; do
  for cfg in "12 4096" "13 2048" "14 1024" "21 16" "22 8" "23 4" "24 2" "24 1"; do
    set -- $cfg
    echo "{\"NTTX_LONGROW\": $mode, \"log_len\": $1, \"batch\": $2}" >> "$out"
    NTTX_LONGROW=$mode timeout -k 10 120 python tools/bench_dist.py --workload batch --log-len $1 --batch $2 --steps 40 --warmup 30 >> "$out" 2>&1 || exit 1
  done
done
