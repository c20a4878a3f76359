#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel + memory-copy trace of an end-to-end run with the device
# front-end (tools/e2e_time.py noto_regular fe), condensed into profiles/<tag>_front_end_summary.md by
# tools/summarize_front_end.py.   usage: tools/profile_front_end.sh <tag>
set -e
TAG=${1:-r2}
WORK=${2:-noto_regular}   # noto_regular | noto_all | many (the 21 fixture files as 21 fonts: two groups in flight)
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
OUT=gpurun_out/prof_${TAG}_front_end
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace -- python3 tools/e2e_time.py $WORK fe > $OUT/e2e.log 2>&1
grep "$WORK" $OUT/e2e.log
