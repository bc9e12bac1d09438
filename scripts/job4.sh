cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
REPO=$PWD
export TMPDIR=/tmp
for cfg in "base:" "r16:--refill-extend 16 --refill-shadow 16" "r16w32:--refill-extend 8208 --refill-shadow 8208" "r8w16:--refill-extend 4104"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  OUT=$REPO/gpurun_out/r03_pmc_$tag
  mkdir -p $OUT
  CMD="python $REPO/bench.py --steps 12 --warmup 16 --no-cpu-baseline --overlap 0 --kernel-timing 0 $args"
  ( cd /tmp
    timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.log 2>&1
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1
  )
  python - <<PY > $REPO/gpurun_out/r03_pmc_$tag.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("k_extend4<false>", "k_shadow4<false", "k_trace4r<false", "k_trace4r<true", "k_commit4"):
            if key in k:
                a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print("==", k)
    for c, a in sorted(acc[k].items()):
        print("   %-28s %.6g   (%d dispatches)" % (c, a[0] / a[1], a[1]))
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(key in r["Name"] for key in ("k_extend4", "k_shadow4", "k_trace4r", "k_commit4", "k_logic", "k_raygen")):
            print("%-110s calls %4s avg %10.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  echo "#### $tag"; cat $REPO/gpurun_out/r03_pmc_$tag.txt
  rm -rf $OUT
done
