#!/bin/bash
# rocprofv3 passes for the round's evidence.  Run on the GPU box: bash tools/profile.sh <tag>
# 1) kernel trace + stats of the bench command; 2)+3) separate PMC passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-extras --dtype ${2:-f16}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
# the same command on ONE stream: with two streams the kernels of the two batches in flight overlap and a
# launch's start..end interval includes waiting for CUs, so it cannot be compared with bench.py's
# per-launch HIP-event durations (taken on one stream)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- $BENCH --streams 1 > $OUT/trace1.log 2>&1
echo "trace1 rc=$?" >> $OUT/trace1.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?" >> $OUT/pmc_fetch.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
echo "write rc=$?" >> $OUT/pmc_write.log
find $OUT -name "*.csv" | head -20
# keep only small files (the merged gpurun_out is capped at 64 MiB)
find $OUT -name "*.csv" -size +20M -delete
du -sh $OUT
