#!/bin/bash
# Everything a round's profiles/ entries are made from, in one call on the GPU box (development aid).
#   tools/round_report.sh <tag>      -> gpurun_out/<tag>_report/*, gpurun_out/prof_<tag>_*
TAG=${1:-r4}
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd "$REPO"
OUT=gpurun_out/${TAG}_report
mkdir -p $OUT
timeout -k 10 300 python3 bench.py > $OUT/bench_noto_regular.json 2> $OUT/bench.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_noto_regular_k20.json 2>> $OUT/bench.err
for w in fira noto_all synthetic; do
  timeout -k 10 300 python3 bench.py --workload $w --no-configs > $OUT/bench_$w.json 2>> $OUT/bench.err
done
# the driver's own command for N > 1 (no launcher: bench.py starts its ranks itself); on a one-GPU box the ranks share device 0
VG_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 2>> $OUT/bench.err | tail -n 1 > $OUT/bench_gpus2_self_launched_one_gpu_rehearsal.json
tools/ubench/clock_probe > $OUT/clock_probe.txt 2>&1
for w in noto_regular noto_all many; do timeout -k 10 120 python3 tools/e2e_sweep.py $w 16 32; done > $OUT/e2e_sweep.txt 2>&1
VGSDF_LIB=versatiles-glyphs-rs_amd/build/dev/libvgsdf.so timeout -k 10 100 python3 tools/kexp.py noto_regular 58 --rep 1 > $OUT/stamps.txt 2>&1
for w in noto_regular fira noto_all; do timeout -k 10 100 python3 tools/kexp.py $w 0 1 --rep 8 --iters 20; done > $OUT/kexp.txt 2>&1
tools/profile.sh ${TAG}_noto > /dev/null 2>&1
tools/profile.sh ${TAG}_synth --workload synthetic > /dev/null 2>&1
tools/profile.sh ${TAG}_fira --workload fira > /dev/null 2>&1
tools/profile.sh ${TAG}_noto_all --workload noto_all > /dev/null 2>&1
tools/profile_front_end.sh ${TAG} > $OUT/front_end.log 2>&1
python3 tools/front_end_timeline.py ${TAG} > $OUT/front_end_timeline.txt 2>&1
for w in noto_regular noto_all many; do timeout -k 10 120 python3 tools/device_span.py $w 60; done > $OUT/device_span.txt 2>&1
timeout -k 10 200 python3 tools/lane_forms.py 2 4 8 > $OUT/lane_forms.txt 2>&1
for n in 2 8; do timeout -k 10 120 python3 tools/stress_lanes.py $n 120 2>&1 | tail -1; done > $OUT/stress_lanes.txt
ls $OUT
