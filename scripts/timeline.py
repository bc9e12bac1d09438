"""Per-iteration timeline of the wavefront step from ONE `rocprofv3 --kernel-trace` pass of bench.py:
    python scripts/timeline.py <kernel_trace.csv> [first_iterations_to_skip] > profiles/rNN_<workload>_timeline.txt
An iteration = [k_logic start, next k_logic start).  For every steady-state iteration: start / end of every kernel relative to the iteration
start, per queue (the main stream's queue and the shadow stream's), the idle time of the main queue (gaps between consecutive kernels on it:
dispatch gaps + waits for the other stream), and the join (main queue idle while only the shadow kernel runs).  Averages at the end."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void flxd::", "").replace("flxd::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r["Queue_Id"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_logic")]
iters = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
iters = iters[skip:]
if not iters:
    sys.exit("no steady-state iterations in the trace")
# the timed window only: bench.py's extra UNTIMED passes behind it (serial schedule, counting variants of the traversal kernels, k_materialise for a
# read-back) are iterations of another shape -- keep the iterations made of the default chain's kernels alone and of ordinary length
import statistics
ODD = ("<true>", "k_extend4<true", "k_shadow4<true", "k_materialise", "k_extend<", "k_shadow<", "rocclr")
std = [(a, b) for a, b in iters if not any(any(o in r[2] for o in ODD) for r in rows[a:b])]
if std:
    med = statistics.median(rows[b][0] - rows[a][0] for a, b in std)
    std = [(a, b) for a, b in std if rows[b][0] - rows[a][0] < 1.25 * med]
    q2 = [sum(1 for r in rows[a:b] if r[3] != rows[a][3]) for a, b in std]          # the two-stream schedule: something runs on the second queue
    std = [ab for ab, n2 in zip(std, q2) if n2 > 0] or std
print(f"# {len(iters)} iterations after the first {skip}; {len(std)} of them are the timed window's (default chain, two streams, no counting / read-back passes)")
iters = std or iters
acc = defaultdict(lambda: [0.0, 0.0, 0.0, 0])       # kernel -> [sum start, sum end, sum dur, n]
tot = defaultdict(float)
shown = 0
for a, b in iters:
    t0 = rows[a][0]; t1 = rows[b][0]
    ks = rows[a:b]
    mainq = ks[0][3]
    main = [k for k in ks if k[3] == mainq]; other = [k for k in ks if k[3] != mainq]
    busy_main = sum(k[1] - k[0] for k in main)
    gaps = []
    prev_end = t0
    for k in main:
        gaps.append((k[0] - prev_end, k[2])); prev_end = max(prev_end, k[1])
    tail_gap = t1 - prev_end                         # after the last main-queue kernel until the next logic starts
    # the part of the main queue's idle time during which a kernel of the other queue was running (join / wait), the rest is dispatch gap
    idle_intervals = []
    prev_end = t0
    for k in main:
        if k[0] > prev_end: idle_intervals.append((prev_end, k[0]))
        prev_end = max(prev_end, k[1])
    if t1 > prev_end: idle_intervals.append((prev_end, t1))
    waited = 0
    for (x, y) in idle_intervals:
        for k in other:
            lo, hi = max(x, k[0]), min(y, k[1])
            if hi > lo: waited += hi - lo
    idle = sum(y - x for x, y in idle_intervals)
    tot["iter"] += t1 - t0; tot["busy_main"] += busy_main; tot["idle_main"] += idle; tot["idle_while_other_runs"] += waited
    tot["other_busy"] += sum(k[1] - k[0] for k in other)
    for k in ks:
        e = acc[(k[2], "main" if k[3] == mainq else "2nd")]
        e[0] += k[0] - t0; e[1] += k[1] - t0; e[2] += k[1] - k[0]; e[3] += 1
    if shown < 2:
        shown += 1
        print(f"--- iteration starting at dispatch {a}: {(t1 - t0) / 1e6:.3f} ms")
        for k in ks:
            print(f"  {'main' if k[3] == mainq else '2nd ':4s} {(k[0] - t0) / 1e6:8.3f} -> {(k[1] - t0) / 1e6:8.3f}  ({(k[1] - k[0]) / 1e6:6.3f} ms)  {k[2][:70]}")
n = len(iters)
print(f"\n=== averages over {n} steady-state iterations (ms)")
print(f"iteration                         {tot['iter'] / n / 1e6:8.3f}")
print(f"main queue busy (sum of kernels)  {tot['busy_main'] / n / 1e6:8.3f}")
print(f"main queue idle                   {tot['idle_main'] / n / 1e6:8.3f}   = waiting while the 2nd queue's kernel runs {tot['idle_while_other_runs'] / n / 1e6:.3f} (the join) + dispatch gaps {(tot['idle_main'] - tot['idle_while_other_runs']) / n / 1e6:.3f}")
print(f"2nd queue busy                    {tot['other_busy'] / n / 1e6:8.3f}")
print(f"check: busy + idle - iteration =  {(tot['busy_main'] + tot['idle_main'] - tot['iter']) / n / 1e6:8.3f}")
print("\nkernel (queue)                                                      start      end      dur   per iteration")
for (name, q), e in sorted(acc.items(), key=lambda kv: kv[1][0] / max(1, kv[1][3])):
    print(f"{name[:58]:58s} {q:4s} {e[0] / e[3] / 1e6:8.3f} {e[1] / e[3] / 1e6:8.3f} {e[2] / e[3] / 1e6:8.3f}   x{e[3] / n:.2f}")
