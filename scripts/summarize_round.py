"""Summarise a scripts/profile_round.sh capture into profiles/<tag>_<workload>_{kernel_stats.csv,counters.txt} and profiles/traffic_<workload>.json
(the file bench.py reads for roofline.traffic / frac; it carries the bench line's `capture_key` incl. the hash of the kernel sources, and
bench.py refuses it as soon as any key differs)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, tag, wl, extra = sys.argv[1], sys.argv[2], sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else "")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(prof, f"{tag}_{wl}_kernel_stats_whole_process.csv"))      # rocprofv3's own summary: every dispatch of the process (settling, extra untimed passes)
KEYS = ("k_trace4r<false", "k_trace4r<true", "k_extend4<false>", "k_shadow4<false", "k_shadow4s<", "k_commit4", "k_lightfix4", "k_extend<false>", "k_shadow<false>", "k_logic", "k_material", "k_raygen",
        "k_queue_scatter", "k_queue_scan")
bench = {}
try:
    bench = json.loads(open(os.path.join(prof, f"{tag}_{wl}_bench.json")).readline())
except Exception:
    pass
# Steady state only: the first `settle_iterations` dispatches of every kernel are the transient from reset (iteration 0 traces nothing but
# primary rays) and are dropped; what is averaged is the timed window + the extra untimed passes over the same steady state.
skip = int(bench.get("settle_iterations", 0))
# ... and (round 5) only the TIMED WINDOW's dispatches: bench.py runs extra untimed passes behind it (serial schedule, counting variants); the kernels of the
# default chain launch once per iteration, so dispatches [skip, skip + steps x windows) of a kernel are the timed ones
timed = int(bench.get("steps", 0)) * int((bench.get("windows") or {}).get("count", 1))
# <tag>_<workload>_kernel_stats.csv (round 6): per-kernel statistics of the TIMED WINDOW only, from the kernel trace -- rocprofv3's own summary averages every
# dispatch of the process, and the launches beside bench.py's 0.5-s counting variants behind the timed window polluted it (k_material_rest "avg 6 ms, max 84 ms").
# The window = [start of the fused logic kernel's dispatch number `skip`, start of its dispatch number `skip + timed`): one logic dispatch per iteration.
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    lg = [r for r in rows if "k_logic" in r[2]]
    if timed and len(lg) > skip + timed:
        t0, t1 = lg[skip][0], lg[skip + timed][0]
        st = collections.defaultdict(list)
        for a, b, k in rows:
            if t0 <= a < t1:
                st[k].append(b - a)
        tot = sum(sum(v) for v in st.values()) or 1
        with open(os.path.join(prof, f"{tag}_{wl}_kernel_stats.csv"), "w") as fo:
            w = csv.writer(fo)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", f"# timed window only: {timed} iterations behind the first {skip}; window {(t1 - t0) / 1e6:.3f} ms = {(t1 - t0) / 1e6 / timed:.4f} ms per iteration"])
            for k, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([k, len(v), sum(v), "%.1f" % (sum(v) / len(v)), "%.2f" % (100.0 * sum(v) / tot), min(v), max(v), ""])
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    rows = collections.defaultdict(list)              # (key, counter) -> [(dispatch id, value)]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in KEYS:
            if key in k:
                rows[(key, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                break
    for (key, cn), lst in rows.items():
        lst.sort()
        lst = lst[skip:] if len(lst) > 2 * skip else lst
        if timed and len(lst) > timed:
            lst = lst[:timed]
        a = acc[key][cn]
        a[0] += sum(v for _, v in lst); a[1] += len(lst)
lines = []
traffic = {}
for k in sorted(acc):
    v = {c: a[0] / a[1] for c, a in acc[k].items()}
    lines.append(f"== {k}   (averages per dispatch, {max(a[1] for a in acc[k].values())} steady-state dispatches of the timed window; the first {skip} of the process and the extra untimed passes dropped)")
    for c in sorted(v):
        lines.append("   %-28s %.6g" % (c, v[c]))
    rd, r32, r64, r128 = (v.get("TCC_EA0_RDREQ_sum", 0), v.get("TCC_EA0_RDREQ_32B_sum", 0), v.get("TCC_EA0_RDREQ_64B_sum", 0), v.get("TCC_EA0_RDREQ_128B_sum", 0))
    wr, w64 = v.get("TCC_EA0_WRREQ_sum", 0), v.get("TCC_EA0_WRREQ_64B_sum", 0)
    if rd or wr:
        rbytes = 32 * r32 + 64 * r64 + 128 * r128 + 64 * max(0.0, rd - r32 - r64 - r128)
        wbytes = 64 * w64 + 32 * (wr - w64)
        lines.append("   fabric read bytes %.6g (requests: 32 B %g, 64 B %g, 128 B %g)   write bytes %.6g   total %.6g" % (rbytes, r32, r64, r128, wbytes, rbytes + wbytes))
        if "FETCH_SIZE" in v:
            lines.append("   cross-check: 2 x FETCH_SIZE + WRITE_SIZE = %.6g bytes (KiB counters; gfx950: FETCH_SIZE tallies 128-B requests at 64 B)" % (2 * v["FETCH_SIZE"] * 1024 + v.get("WRITE_SIZE", 0) * 1024))
        traffic[k] = {"read_bytes": rbytes, "write_bytes": wbytes, "read_requests_128B": r128, "read_requests_64B": r64, "read_requests_32B": r32,
                      "write_requests_64B": w64, "write_requests_32B": wr - w64, "fetch_size_kib": v.get("FETCH_SIZE"), "write_size_kib": v.get("WRITE_SIZE")}
    if v.get("SQ_INSTS_VALU"):
        lines.append("   lanes active per VALU instruction %.1f of 64" % (v.get("SQ_THREAD_CYCLES_VALU", 0) / v["SQ_INSTS_VALU"]))
open(os.path.join(prof, f"{tag}_{wl}_counters.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
ext = next((k for k in ("k_trace4r<false", "k_extend4<false>", "k_extend<false>") if k in traffic), None)
if ext:
    cfg = bench.get("config", {})
    t = traffic[ext]
    extra_k = {k: traffic[k]["read_bytes"] + traffic[k]["write_bytes"] for k in ("k_commit4", "k_lightfix4") if k in traffic}
    j = {"source": f"scripts/profile_round.sh {tag} {wl} {extra}".strip() + " -> profiles/%s_%s_counters.txt (rocprofv3 --pmc, separate passes with --kernel-trace only)" % (tag, wl),
         "capture_key": (bench.get("roofline") or {}).get("capture_key"),
         "kernel": ext, "workload": wl, "extend_tree": 4 if "4" in ext else 2, "num_tasks": cfg.get("num_tasks_per_gpu"), "refill_extend": cfg.get("refill_extend", 0),
         "extend_read_requests_128B": t["read_requests_128B"], "extend_read_requests_64B": t["read_requests_64B"], "extend_write_requests_64B": t["write_requests_64B"],
         "extend_write_requests_32B": t["write_requests_32B"], "fetch_size_kib": t["fetch_size_kib"], "write_size_kib": t["write_size_kib"],
         "companion_kernels_bytes": extra_k,
         "extend_valu_instructions": acc[ext].get("SQ_INSTS_VALU", [0, 1])[0] / max(1, acc[ext].get("SQ_INSTS_VALU", [0, 1])[1]) or None,
         "extend_lanes_per_valu_instruction": (acc[ext]["SQ_THREAD_CYCLES_VALU"][0] / acc[ext]["SQ_INSTS_VALU"][0]) if acc[ext].get("SQ_INSTS_VALU", [0])[0] else None,
         "correction": "gfx950 guide: FETCH_SIZE = TCC_EA0_RDREQ x 64 B although the requests are 128 B -> read bytes = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B; write bytes = 64 x WRREQ_64B + 32 x the rest.  Fabric-side counts: Infinity-Cache hits are included, HBM proper is lower.",
         "extend_hbm_bytes_per_launch": t["read_bytes"] + t["write_bytes"] + sum(extra_k.values())}

    def per_launch(keys):
        """fabric bytes / VALU instructions / lanes per instruction of the dispatches of `keys` that make up ONE launch of the pass (e.g. the fused
        logic pass = k_logic + k_queue_scan + k_queue_scatter: flx_wf_logic's timer covers all three)"""
        ks = [k for k in keys if k in acc]
        if not ks:
            return None
        by = sum(traffic[k]["read_bytes"] + traffic[k]["write_bytes"] for k in ks if k in traffic) or None
        vi = sum(acc[k]["SQ_INSTS_VALU"][0] / max(1, acc[k]["SQ_INSTS_VALU"][1]) for k in ks if acc[k].get("SQ_INSTS_VALU", [0])[0])
        tc = sum(acc[k]["SQ_THREAD_CYCLES_VALU"][0] / max(1, acc[k]["SQ_THREAD_CYCLES_VALU"][1]) for k in ks if acc[k].get("SQ_THREAD_CYCLES_VALU", [0])[0])
        return {"kernels": ks, "hbm_bytes_per_launch": by, "valu_instructions_per_launch": vi or None, "lanes_per_valu_instruction": (tc / vi) if vi else None}
    j["passes"] = {"extend": per_launch([ext]), "logic": per_launch(["k_logic", "k_queue_scan", "k_queue_scatter"]),
                   "shadow": per_launch(["k_shadow4<false", "k_shadow4s<", "k_trace4r<true", "k_shadow<false>", "k_lightfix4"])}
    json.dump(j, open(os.path.join(prof, f"traffic_{wl}.json"), "w"), indent=1)
    print("wrote traffic_%s.json: %.4g bytes per launch" % (wl, j["extend_hbm_bytes_per_launch"]))
