#!/bin/bash
# PMC counters (VALU instructions / busy cycles / wave cycles) of kernel variants on a resident batch
# (development aid).  usage: tools/pmc_kexp.sh <workload> <variant>...   (set VGSDF_LIB for dev variants)
W=$1; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
for V in "$@"; do
  OUT=gpurun_out/pmck_${W}_v$V
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 tools/kexp.py $W $V --rep 1 --iters 2 > $OUT/a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/b -- python3 tools/kexp.py $W $V --rep 1 --iters 2 > $OUT/b.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sdf_tiles" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant $V ($W): " + "  ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(acc.items())))
PY
done
