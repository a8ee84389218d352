#!/bin/bash
# socket power and clocks while the headline bench loop runs (rocm-smi, sampled every 1.5 s)
timeout -k 10 100 python bench.py --no-cpu --no-context --steps 30000 --warmup 50 > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
pid=$!
for i in $(seq 1 12); do
  sleep 1.5
  echo "t=$i $(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i -E "Power \(W\)|sclk|junction" | sed 's/.*: //' | tr '\n' ' ')"
done
wait $pid
grep -o '"ms_per_step": [0-9.]*' gpurun_out/power_bench.json
