#!/bin/bash
# Column kernels built with a floor on waves per SIMD (register budget): measurement builds under tools/ablate/
#   occA: split-table passes of 2^8 / 2^9 points at >= 4 waves (<= 128 registers)    occB: ... and 2^7 points at >= 6 (<= 80)
#   occC / occD: plain 2^7-point passes (products of 2^16 points) at >= 8 / 7 waves (<= 64 / 72 registers)
run() { NTTX_LIB=$PWD/$1 timeout -k 10 100 python bench.py "${@:2}" --no-cpu --no-context 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us', d['config'].get('passes'), {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1; }
for lib in ntt_amd/csrc/libnttx.so tools/ablate/libnttx_occA.so tools/ablate/libnttx_occB.so; do
  for k in 24 25 26; do echo "== $lib 2^$k"; run $lib --workload fourstep --log-len $k --steps 40 --warmup 10; done
done
for lib in ntt_amd/csrc/libnttx.so tools/ablate/libnttx_occC.so tools/ablate/libnttx_occD.so; do
  echo "== $lib batch 2^16 x 4096"; run $lib --workload batch --steps 20 --warmup 5
done
