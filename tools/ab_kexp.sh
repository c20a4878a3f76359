#!/bin/bash
# A/B of experimental builds of the raster kernel on one box (development aid).
#   tools/ab_kexp.sh <workload> <lib>...     lib = path of a libvgsdf build, "" = the product library
# Per library: parity against the golden SHA-256s + us per launch / per replica (tools/kexp.py), then the VALU / wait
# counters of the span kernel from two rocprofv3 --pmc passes.
W=$1; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
for L in "$@"; do
  TAG=$(basename "${L:-product}" .so)
  export VGSDF_LIB=$L
  [ -z "$L" ] && unset VGSDF_LIB
  echo "== $TAG"
  python3 tools/kexp.py $W 0 --rep 8 2>&1 | grep variant
  OUT=gpurun_out/ab_$TAG
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 tools/kexp.py $W 0 --rep 1 --iters 2 > $OUT/a.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/b -- python3 tools/kexp.py $W 0 --rep 1 --iters 2 > $OUT/b.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sdf_tiles" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   " + "  ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(acc.items())))
PY
done
