#!/bin/bash
# SQ counters of EVERY kernel of a bench run (serial schedule unless the args say otherwise): scripts/pmc2.sh <tag> [bench args] -> gpurun_out/pmc2_<tag>.txt
TAG=${1:-x}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/pmc2_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 16 --windows 1 --no-cpu-baseline --kernel-timing 0 $*"
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
cd $REPO
python - > gpurun_out/pmc2_$TAG.txt <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void flxd::", "")
        if "rocclr" in k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void flxd::", "")
        d = dur[k]; d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
print("# $TAG: bench.py $*  (averages per dispatch over the whole run, counting passes included)")
for k in sorted(acc):
    c = {n: a[0] / a[1] for n, a in acc[k].items()}
    n = max(a[1] for a in acc[k].values())
    lanes = c.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, c.get("SQ_INSTS_VALU", 1)) if "SQ_INSTS_VALU" in c else 0
    print("== %-40s dispatches %4d  avg %8.1f us (under the counters)  VALU insts %.4g  lanes/inst %.1f  waves %.4g  VMEM_RD %.4g  LDS %.4g  SALU %.4g" % (
        k[:40], n, dur[k][0] / max(1, dur[k][1]), c.get("SQ_INSTS_VALU", 0), lanes / 1.0, c.get("SQ_WAVES", 0), c.get("SQ_INSTS_VMEM_RD", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_SALU", 0)))
    print("      busy_cycles %.4g  wave_cycles %.4g  wait_any %.4g  wait_inst_any %.4g  active_inst_any %.4g  active_inst_valu %.4g  branch %.4g" % (
        c.get("SQ_BUSY_CYCLES", 0), c.get("SQ_WAVE_CYCLES", 0), c.get("SQ_WAIT_ANY", 0), c.get("SQ_WAIT_INST_ANY", 0), c.get("SQ_ACTIVE_INST_ANY", 0), c.get("SQ_ACTIVE_INST_VALU", 0), c.get("SQ_INSTS_BRANCH", 0)))
PY
cat gpurun_out/pmc2_$TAG.txt | grep -A1 "shadow\|trace4r"
