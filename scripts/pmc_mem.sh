#!/bin/bash
# memory-path PMC passes for the trace kernels (L1->L2 latency, L2 hit/miss, fabric reads, TLB): scripts/pmc_mem.sh <tag> [bench args]
TAG=${1:-x}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/pmcmem_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 16 --no-cpu-baseline $*"
cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("k_extend<false>", "k_shadow<false>", "k_trace_stream", "k_logic", "k_material<1>"):   # <false> = the product kernels, not the STATS counting variants
            if key in k:
                a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print("==", k)
    for c, a in sorted(acc[k].items()):
        print("   %-44s %.6g" % (c, a[0] / a[1]))
PY
