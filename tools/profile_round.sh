#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the un-graphed bench (kernels launched from hipGraphs are not listed by --kernel-trace)
#   2. HBM traffic counters in SEPARATE passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes
# Outputs land in gpurun_out/prof_$1/ ; copy the summaries into profiles/ afterwards.
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-ddim --no-graph"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1; echo trace_rc=$?
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1; echo fetch_rc=$?
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1; echo write_rc=$?
ls -la $OUT/*/* | head -30
