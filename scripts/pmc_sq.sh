#!/bin/bash
# SQ / VALU occupancy counters of the traversal kernels, serial schedule: scripts/pmc_sq.sh <tag> [bench args]
TAG=${1:-x}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/pmcsq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 16 --no-cpu-baseline --overlap 0 --kernel-timing 0 $*"
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("k_extend4<false>", "k_shadow4<false", "k_extend<false>", "k_shadow<false>", "k_logic", "k_material<1>", "k_raygen"):
            if key in k:
                a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print("==", k)
    for c, a in sorted(acc[k].items()):
        print("   %-44s %.6g   (%d dispatches)" % (c, a[0] / a[1], a[1]))
PY
