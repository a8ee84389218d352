#!/bin/bash
# Power, power cap, throttle status and shader clock while the headline step runs back to back (about 12 s).
# usage: tools/power_trace.sh [bench.py arguments]        (GPU box)
out=${POWER_OUT:-gpurun_out/power_trace.txt}
{ rocm-smi --showmaxpower --showclocks; amd-smi static --limit 2>&1 | head -40; } > "$out.idle" 2>&1
python bench.py --no-cpu --no-context --steps 20000 --warmup 50 "$@" > "$out.bench" 2>/dev/null &
pid=$!
sleep 4
: > "$out"
for i in $(seq 1 6); do
  echo "--- sample $i" >> "$out"
  amd-smi metric --power --clock --temperature --usage 2>&1 | grep -E -i "SOCKET_POWER|THROTTLE|GFX_ACTIVITY|CLK:|GFX_0:|EDGE|HOTSPOT|MEM:|POWER|clk_locked|deep_sleep" | head -30 >> "$out"
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | tr -s ' ' >> "$out"
  sleep 1
done
wait $pid
python -c "
import json,sys; d=json.loads(open('$out.bench').read()); print('bench:', round(d['value']), 'products/s,', round(d['ms_per_step']*1e3,1), 'us per step')" >> "$out"
