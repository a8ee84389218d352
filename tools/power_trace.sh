#!/bin/bash
# Socket power and shader clocks while a workload runs back to back for about ten seconds (GPU box).
# usage: tools/power_trace.sh <label> <command ...>     e.g.  tools/power_trace.sh single python bench.py --no-cpu --no-context --steps 20000 --warmup 50
# Writes gpurun_out/power_<label>.txt: six samples of `amd-smi metric --power --clock` taken from the fourth second on, then
# the workload's own output line.
label=$1; shift
out=gpurun_out/power_$label.txt
mkdir -p gpurun_out
"$@" > "$out.run" 2>/dev/null &
pid=$!
sleep 4
{ echo "# $*"; rocm-smi --showmaxpower 2>/dev/null | grep -E "Power \(W\)" | tr -s ' \t' ' '; } > "$out"
for i in $(seq 1 6); do
  amd-smi metric --power --clock 2>&1 | python3 -c "
import re,sys
t=sys.stdin.read()
pw=re.findall(r'SOCKET_POWER: (\d+) W',t)
gfx=t.split('GFX_0:')[1].split('MEM_0:')[0] if 'GFX_0:' in t and 'MEM_0:' in t else t
clk=[int(x) for x in re.findall(r'^\s+CLK: (\d+) MHz',gfx,re.M)]
print('socket power', '/'.join(pw) or 'n/a', 'W; gfx clocks MHz', clk)" >> "$out"
  sleep 1
done
wait $pid
tail -n 1 "$out.run" | cut -c1-400 >> "$out"
rm -f "$out.run"
