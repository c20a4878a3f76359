#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command.
# usage: tools/profile.sh <tag> [bench args...]    -> gpurun_out/prof_<tag>/{trace,fetch,write,sq}
set -e
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-e2e --no-configs --no-two-in-flight $*"  # >= 50 ms of launches: the average is not a cold one
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
# PMC passes: counters only (never combined with other trace domains on this pool)
PMC="python3 bench.py --steps 3 --warmup 1 --min-ms 0 --no-cpu-baseline --no-e2e --no-configs --no-two-in-flight $*"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $PMC > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $PMC > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $PMC > $OUT/sq.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -- $PMC > $OUT/sq2.log 2>&1 || true
# VALU instruction classes (issue-cost model of the VALU roofline)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq3 -- $PMC > $OUT/sq3.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/sq4 -- $PMC > $OUT/sq4.log 2>&1 || true
find $OUT -name "*.csv" | tail -40
