#!/bin/bash
# rocprofv3 passes over one benchmark workload: kernel trace + counter passes (each its own run).
# usage: tools/prof.sh <tag> [bench.py arguments, e.g. --workload crt]     (on the GPU box; results under gpurun_out/prof_<tag>)
# then:  python3 tools/summarize_prof.py <tag> <counter_record_key of that bench line> "<the bench.py arguments>"
set -e
tag=${1:-r1}; shift || true
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py $* --no-cpu --no-context"
echo "$BENCH" > $out/command.txt
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH > $out/trace.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $BENCH > $out/pmc_fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $BENCH > $out/pmc_write.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq -- $BENCH > $out/pmc_sq.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $out/pmc_sq2 -- $BENCH > $out/pmc_sq2.log 2>&1
grep -h '"metric"' $out/trace.log | tail -1 > $out/bench_line.json || true
find $out -name "*.csv" | head -40
