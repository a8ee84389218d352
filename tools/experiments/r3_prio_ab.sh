#!/bin/bash
# Wave priority inside the column kernels (measurement builds with -DNTTX_PRIO=1: butterfly rounds at priority 2;
# -DNTTX_PRIO=2: loads, factor multiplications and stores at priority 2), headline workload, same box, two rounds
run() { NTTX_LIB=$PWD/$1 timeout -k 10 100 python bench.py --no-cpu --no-context --steps 200 --warmup 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us', {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" || exit 1; }
for rep in 1 2; do
  for lib in ntt_amd/csrc/libnttx.so tools/ablate/libnttx_prio1.so tools/ablate/libnttx_prio2.so; do echo "== $lib"; run $lib; done
done
