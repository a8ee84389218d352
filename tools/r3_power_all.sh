#!/bin/bash
# power traces of the five configurations (about 15 s each)
tools/power_trace.sh single python bench.py --no-cpu --no-context --steps 20000 --warmup 50 &&
tools/power_trace.sh crt python bench.py --workload crt --no-cpu --no-context --steps 25000 --warmup 50 &&
tools/power_trace.sh batch python bench.py --workload batch --no-cpu --no-context --steps 5000 --warmup 20 &&
tools/power_trace.sh fourstep26 python bench.py --workload fourstep --log-len 26 --no-cpu --no-context --steps 10000 --warmup 20 &&
STEPS=15000 tools/power_trace.sh direct64 python tools/bench_direct64.py 18:128
