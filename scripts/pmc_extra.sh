#!/bin/bash
# extra PMC passes (TA/TCP/SQ detail) for the trace kernels: scripts/pmc_extra.sh <tag> [bench args]
TAG=${1:-x}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 16 --no-cpu-baseline $*"
cd /tmp
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("k_extend<false>", "k_shadow<false>", "k_trace_stream", "k_logic", "k_material<1>"):
            if key in k:
                a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print("==", k)
    for c, a in sorted(acc[k].items()):
        print("   %-44s %.5g" % (c, a[0] / a[1]))
PY
