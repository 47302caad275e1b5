#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the un-graphed bench (kernels launched from hipGraphs are not listed by --kernel-trace)
#   2. HBM traffic counters in SEPARATE passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes
#   3. MFMA utilisation counters in their own pass (SQ_VALU_MFMA_BUSY_CYCLES against the busy / elapsed cycles)
# usage: tools/profile_round.sh <tag> [workload]      outputs: gpurun_out/prof_<tag>/ ; tools/summarize_profile.py <tag>
# copies the summaries into profiles/.  The program sits directly after `--` (no env / bash -c hop: rocprofv3's preloaded
# library has initialised the GPU by then).
set -o pipefail
TAG=${1:-r05_tr}
WL=${2:-transformer_denoiser_T50}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/csrc_hash.py > $OUT/csrc_hash.txt     # the build these counters belong to
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $WL --steps 40 --warmup 5 --no-cpu-baseline --no-ddim --no-mlp --no-cli-path --no-graph"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1; echo trace_rc=$?
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1; echo fetch_rc=$?
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1; echo write_rc=$?
timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_mfma.log 2>&1; echo mfma_rc=$?
ls -la $OUT/*/* | head -40
