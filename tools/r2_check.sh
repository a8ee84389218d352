#!/bin/bash
# check on the GPU box: tests, headline bench, the other workloads on one rank, two-rank rehearsals on one GPU
set -o pipefail
out=gpurun_out/r2a; mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -3 $out/tests.log
timeout -k 10 200 python bench.py > $out/bench_single.json 2> $out/bench_single.err || { tail $out/bench_single.err; exit 1; }
for w in crt batch fourstep; do
  timeout -k 10 200 python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err || { tail $out/bench_$w.err; exit 1; }
done
for w in single crt batch fourstep; do
  NTTX_BENCH_ONE_DEVICE=1 timeout -k 10 200 python bench.py --gpus 2 --workload $w --steps 5 --warmup 2 --no-cpu > $out/reh2_$w.json 2> $out/reh2_$w.err || { tail $out/reh2_$w.err; exit 1; }
done
NTTX_BENCH_ONE_DEVICE=1 timeout -k 10 200 python bench.py --gpus 4 --workload crt --steps 3 --warmup 1 --no-cpu > $out/reh4_crt.json 2> $out/reh4_crt.err || { tail $out/reh4_crt.err; exit 1; }
NTTX_BENCH_ONE_DEVICE=1 timeout -k 10 200 python bench.py --gpus 4 --workload fourstep --steps 3 --warmup 1 --no-cpu > $out/reh4_fourstep.json 2> $out/reh4_fourstep.err || { tail $out/reh4_fourstep.err; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 120 python tools/lat1.py 18 300 > $out/lat1.log 2>&1 || { tail $out/lat1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $out/lat1_trace -- python3 tools/lat1.py 18 300 > $out/lat1_trace.log 2>&1 || { tail $out/lat1_trace.log; exit 1; }
python tools/trace_gaps.py $out/lat1_trace 600 > $out/lat1_gaps.txt
cat $out/lat1.log $out/lat1_gaps.txt
for f in $out/bench_*.json $out/reh*.json; do echo "== $f"; cut -c1-400 $f; done
