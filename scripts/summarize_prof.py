"""Summarise a scripts/profile.sh capture: per-kernel time (kernel-trace) and per-kernel counter averages."""
import csv, glob, os, sys, collections
out = sys.argv[1]


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def short(name):
    n = name.split("(")[0]
    for k in ("k_extend4", "k_shadow4", "k_extend", "k_shadow", "k_logic", "k_material", "k_raygen", "k_reset", "k_queue_scan", "k_queue_scatter",
              "k_end_iteration", "k_postprocess", "k_state"):
        if k in n:
            if k == "k_material":
                return n[n.find("k_material"):][:24]
            if k == "k_logic":
                return n[n.find("k_logic"):][:24]          # k_logic<0> plain | <1> diffuse step inline | <31> every BSDF inline
            if k in ("k_extend4", "k_shadow4", "k_extend", "k_shadow"):
                return k + ("<STATS>" if "<true" in name or "ILb1E" in name else "")
            return k
    return n[:40]


f = find("trace", "*kernel_stats.csv")
if f:
    print("== kernel time (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print("  %-28s calls %6s  total %10.3f ms  avg %9.3f us  %5s%%" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                     float(r["AverageNs"]) / 1e3, r["Percentage"]))
f = find("trace", "*kernel_trace.csv")
if f:
    # the counting (<STATS>) variants run for tens of ms on one stream while the other stream's kernels of the same step crawl beside
    # them, which drags the plain averages above: the product figures = dispatches that do not overlap a <STATS> dispatch in time
    rows = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
    stats = sorted((a, b) for k, a, b in rows if k.endswith("<STATS>"))
    per = collections.defaultdict(list)
    for k, a, b in rows:
        if k.endswith("<STATS>") or any(a < sb and sa < b for sa, sb in stats):
            continue
        per[k].append((b - a) / 1e3)
    print("== product dispatches only (not overlapping a <STATS> counting pass), from", os.path.relpath(f, out))
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        print("  %-28s calls %6d  avg %9.3f us  median %9.3f us  min %9.3f  max %9.3f" % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[0], v[-1]))
for sub in ("pmc_fetch", "pmc_write", "pmc_tcc", "pmc_sq"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        print("==", sub, ": no counter_collection.csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); c = r["Counter_Name"]; v = float(r["Counter_Value"])
        a = acc[k][c]; a[0] += v; a[1] += 1
    print("==", sub, "(average per dispatch)")
    for k in sorted(acc):
        print("  %-28s %s" % (k, "  ".join("%s=%.4g" % (c, a[0] / a[1]) for c, a in sorted(acc[k].items()))))
