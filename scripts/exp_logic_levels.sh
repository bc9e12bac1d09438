#!/bin/bash
# The fused logic pass lands on different time levels from one PROCESS to the next (round 3: 0.46 / 0.48 / 0.51 ms at 4 M paths).  N processes of the
# same short serial-schedule bench under rocprofv3 with the L1 TLB and DRAM-request counters: per process the pass's average duration (kernel trace)
# beside TCP_UTCL1 requests / misses and the L2's DRAM read requests.   usage: scripts/exp_logic_levels.sh [N] > profiles/rNN_logic_levels.txt
N=${1:-8}
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
for i in $(seq 1 $N); do
  rm -rf /tmp/lvl_$i
  timeout 600 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/lvl_$i -- python $REPO/bench.py --steps 10 --warmup 18 --windows 1 --no-cpu-baseline --overlap 0 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
dur = []; cnt = collections.defaultdict(list)
for f in glob.glob("/tmp/lvl_$i/**/*counter_collection.csv", recursive=True):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_logic<1, true>" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[int(r["Dispatch_Id"])]["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(rows)[18:]
    for d in ids:
        dur.append(rows[d]["dur"])
        for k, v in rows[d].items():
            if k != "dur": cnt[k].append(v)
if dur:
    print("process $i: k_logic<1,true> %.1f us avg over %d launches (min %.1f max %.1f) | " % (sum(dur) / len(dur), len(dur), min(dur), max(dur)) + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(cnt.items())))
else:
    print("process $i: no data")
PY
done
