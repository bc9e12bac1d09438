#!/bin/bash
# L2 (TCC) input-side PMC passes for the trace kernels: scripts/pmc_tcc.sh <tag> [bench args]
TAG=${1:-x}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/pmctcc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 16 --no-cpu-baseline $*"
cd /tmp
i=0
for set in "TCC_IB_STALL_sum TCC_IB_REQ_sum" "TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum" "TCC_BUBBLE_sum TCC_CYCLE_sum" "TCC_TAG_STALL_sum TCC_REQ_sum" \
           "TCC_READ_sum TCC_WRITE_sum" "TCC_NC_REQ_sum TCC_UC_REQ_sum" "TCC_CC_REQ_sum TCC_RW_REQ_sum" "TCC_STREAMING_REQ_sum TCC_PROBE_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("k_extend<false>", "k_shadow<false>", "k_logic", "k_material<1>"):
            if key in k:
                a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print("==", k)
    for c, a in sorted(acc[k].items()):
        print("   %-44s %.6g" % (c, a[0] / a[1]))
PY
