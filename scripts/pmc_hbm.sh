#!/bin/bash
# exact fabric-side byte count of the trace kernels: request counts by size (scripts/pmc_hbm.sh <tag> [bench args])
TAG=${1:-x}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/pmchbm_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 16 --no-cpu-baseline $*"
cd /tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("k_extend4<false>", "k_shadow4<false", "k_extend<false>", "k_shadow<false>", "k_logic", "k_material<1>", "k_material_rest", "k_raygen", "k_queue_scatter"):
            if key in k:
                a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print("==", k)
    v = {c: a[0] / a[1] for c, a in acc[k].items()}
    for c in sorted(v):
        print("   %-32s %.6g" % (c, v[c]))
    rd = v.get("TCC_EA0_RDREQ_sum", 0); r32 = v.get("TCC_EA0_RDREQ_32B_sum", 0); r64 = v.get("TCC_EA0_RDREQ_64B_sum", 0); r128 = v.get("TCC_EA0_RDREQ_128B_sum", 0)
    wr = v.get("TCC_EA0_WRREQ_sum", 0); w64 = v.get("TCC_EA0_WRREQ_64B_sum", 0)
    print("   read bytes by request size   %.6g  (32B %g, 64B %g, 128B %g; other %g)" % (32 * r32 + 64 * r64 + 128 * r128, r32, r64, r128, rd - r32 - r64 - r128))
    print("   write bytes (64B x WRREQ_64B + 32B x rest)  %.6g" % (64 * w64 + 32 * (wr - w64)))
PY
