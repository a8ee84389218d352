#!/bin/bash
# Long-row fused products of p = 469762049: signed four-multiple lanes (default) against the unsigned nine-multiple class
# (NTTX_UNSIGNED=1, which also switches the 512-point kernel: only the long-row lengths are measured here).
out=gpurun_out/r3e; mkdir -p $out; rm -f $out/longrow_ab.txt
for u in 0 1; do
  echo "== NTTX_UNSIGNED=$u" >> $out/longrow_ab.txt
  for k in 12 13 14; do
    NTTX_UNSIGNED=$u timeout -k 10 100 python tools/bench_dist.py --workload batch --log-len $k --batch $((1<<(26-k))) --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2^$k', round(d['us_per_mult_per_gpu'],4), 'us each', d['kernels_ms_per_step'], d['passes'])" >> $out/longrow_ab.txt || exit 1
  done
  for k in 21 22 23; do
    NTTX_UNSIGNED=$u timeout -k 10 100 python bench.py --workload fourstep --log-len $k --no-cpu --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2^$k', round(d['ms_per_step']*1e3,1), 'us', d['config']['passes'], {k: round(v*1e3,1) for k,v in d['kernels_ms_per_launch'].items()})" >> $out/longrow_ab.txt || exit 1
  done
done
cat $out/longrow_ab.txt
