#!/usr/bin/env python3
"""The VALU-issue roof of the hot kernels, computed from the ISA of the SHIPPED code object instead of a 2 ... 4-cycle bracket.

    python scripts/valu_roof.py [--lib fluctus_amd/libfluctus_hip.so] [--out profiles/valu_roof.json]

Round 5's bench line priced a kernel's SQ_INSTS_VALU at "4 cycles per wave64 instruction" (every instruction on the SIMD's one single-issue pipe)
with "2" (every instruction dual-issuable) beside it; the all-types logic pass then showed an issue fraction of 1.13 -- the 4 was not the cost of
that mix.  This tool disassembles the gfx950 code objects inside libfluctus_hip.so (llvm-objdump --offloading / -d), finds each hot kernel's loops
(back edges of the branch graph), classifies every VALU instruction by the issue class scripts/ubench/valu_rate.hip / valu_pairs*.hip measured
(profiles/r03_ubench_valu_rate.txt, r03_ubench_valu_pairs.txt, r03_ubench_valu_pairs_membership.txt) and prices a region with a two-pipe model:

  class  pipe time  instructions (wave64, SIMD cycles, 4-8 waves per SIMD)
  F      4.4, either pipe   v_mov, v_mul_f32, v_add/sub_f32, v_and/or/xor_b32, v_add/sub_u32, v_fmac_f32, v_fmamk/fmaak_f32        (2.1-2.4 alone, 4.2-4.6 per pair)
  M      5.5, either pipe   v_fma_f32 (three VGPR sources)                                                                       (2.7-2.9 alone, 5.2-5.8 per fma + F pair)
  S      4.2, pipe A only   conversions, min / max / med3, compares, v_cndmask, shifts, bfe / bfi / perm, and_or / add3 / lshl_add, 24-bit and 32-bit integer
                            multiplies, v_fma_mix, SDWA / DPP forms, lane reads, the v_div_* helpers                              (4.0-4.4 alone, 8.1-8.6 per S + S pair, 4.2-4.8 per S + F pair)
  T      8.2, pipe A only   v_rcp / rsq / sqrt / exp / log / sin / cos_f32                                                        (8.1-8.3 alone)
  P      5.9, pipe A only   v_pk_*                                                                                                 (5.8-6.2 alone)
  cycles(region) = max(A-only work, (all work) / 2)          -- pipe A takes everything, pipe B only classes F and M; no dependency stalls: a ROOF.

It reproduces the measured pairs (S + S 8.4, S + F 4.3, F + F 4.4, fma + cvt 4.85 vs 4.2-4.3 measured, fma + cndmask 4.85 vs 5.1-5.9).  A kernel's
figure is the trip-weighted mean over its regions (weights below: wave-level trip counts of the lab build, profiles/r04_lane_use.txt, or one trip per
instruction for the straight-line logic pass); `range` = the cheapest and the dearest region, i.e. what ANY weighting could give.  bench.py multiplies
SQ_INSTS_VALU of the PMC capture by `cycles_per_instruction` (only while source_hash matches) for roofline.valu.issue_frac.
Nothing here is imported by the product or the tests."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

F_OPS = {"v_mov_b32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
         "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mac_f32", "v_mul_legacy_f32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_accvgpr_mov_b32"}
M_OPS = {"v_fma_f32", "v_mad_f32", "v_fma_legacy_f32"}
T_OPS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_exp_legacy_f32", "v_log_legacy_f32"}
COST = {"F": 4.4, "M": 5.5, "S": 4.2, "T": 8.2, "P": 5.9}
# instructions the micro-benchmarks did not time: priced as class S (pipe A only: the dearer assumption); their share is reported as `unmeasured_frac`
MEASURED_S = ("v_cvt_", "v_max", "v_min", "v_med3", "v_cmp", "v_cndmask", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_bfi", "v_perm", "v_and_or", "v_add3", "v_mad_u32_u24",
              "v_mul_u32_u24", "v_mul_lo_u32", "v_fma_mix", "v_mad_mix")


def classify(mn):
    """mnemonic (encoding suffix included) -> (class, measured?)"""
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", mn)
    mod = mn.endswith("_sdwa") or mn.endswith("_dpp")
    if base.startswith("v_pk_"):
        return "P", True
    if base in T_OPS:
        return "T", True
    if mod:
        return "S", True                                    # (sdwa_add: 4.1-4.8 alone; DPP forms share the operand path)
    if base in M_OPS:
        return "M", True
    if base in F_OPS:
        return "F", base not in ("v_mac_f32", "v_mul_legacy_f32") and not base.startswith("v_accvgpr")
    return "S", base.startswith(MEASURED_S)


def cycles(hist):
    """two-pipe model: total SIMD cycles for a histogram {class: count}"""
    a_only = sum(COST[c] * hist.get(c, 0) for c in ("S", "T", "P"))
    either = sum(COST[c] * hist.get(c, 0) for c in ("F", "M"))
    return max(a_only, (a_only + either) / 2.0)


def extract(lib, tmp):
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    out = {}
    for f in sorted(os.listdir(tmp)):
        if "gfx950" not in f:
            continue
        txt = subprocess.run([OBJDUMP, "-d", f], cwd=tmp, stdout=subprocess.PIPE, text=True, check=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", line)
            if m:
                cur = m.group(2); out[cur] = {"start": int(m.group(1), 16), "ins": []}
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-Fa-f]+):", line)
            if m and cur:
                tgt = re.search(r"<[^>]*\+0x([0-9a-f]+)>\s*$", line)
                out[cur]["ins"].append((int(m.group(3), 16), m.group(1), (out[cur]["start"] + int(tgt.group(1), 16)) if tgt else None))
    return out


def demangle_short(sym):
    r = subprocess.run(["c++filt", sym], stdout=subprocess.PIPE, text=True).stdout.strip()
    return re.sub(r"\(.*$", "", r).replace("void ", "").replace("flxd::", "")


def loops_of(ins):
    """natural loops from back edges (branch to a lower address): [(lo, hi)] address ranges, innermost first"""
    rng = sorted({(t, a) for a, mn, t in ins if t is not None and t <= a and "branch" in mn})
    merged = {}
    for lo, hi in rng:                                      # several back edges to one header = one loop
        merged[lo] = max(merged.get(lo, lo), hi)
    loops = sorted(merged.items(), key=lambda r: r[1] - r[0])
    return loops


def hist_of(ins, lo=None, hi=None, exclude=()):
    h, ops, unmeasured = Counter(), Counter(), 0
    for a, mn, _ in ins:
        if not mn.startswith("v_") or mn.startswith("v_nop"):
            continue
        if lo is not None and not (lo <= a <= hi):
            continue
        if any(l <= a <= h_ for l, h_ in exclude):
            continue
        c, meas = classify(mn)
        h[c] += 1; ops[re.sub(r"_(e32|e64)$", "", mn)] += 1
        unmeasured += 0 if meas else 1
    return h, ops, unmeasured


def region_report(name, h, ops, unmeasured, trips):
    n = sum(h.values())
    return {"region": name, "valu_instructions": n, "classes": dict(h), "cycles_per_instruction": (cycles(h) / n) if n else None, "trips_weight": trips,
            "unmeasured_frac": (unmeasured / n) if n else 0.0, "top_opcodes": dict(ops.most_common(12))}


# wave-level trip counts per 64-ray block of the persistent closest-hit kernel (lab build -DFLX_LAB_RSTATS, profiles/r04_lane_use.txt: kitchen / conference
# 18.3 / 18.6 node-visit rounds, 3.0 / 2.5 leaf phases, 11.6 / 9.3 triangle-loop iterations); the thread-per-ray any-hit kernel has no such lab counters:
# per-ray visit counts of its counting variant (7.3 node visits, 0.9 leaves, 3.0 triangle tests on the kitchen) stand in -- only the RATIO matters, and
# `range` brackets every possible ratio.
TRIPS = {"trace": {"node": 18.4, "tri": 10.5, "other": 3.0}, "shadow": {"node": 7.3, "tri": 3.0, "other": 1.0}}


def traversal_kernel(ins, kind):
    """regions of a traversal kernel: the NODE-VISIT loop = the smallest loop whose own instructions (nested loops excluded) hold the 24 byte -> float
    conversions of a 4-wide node test; the TRIANGLE loop = the smallest loop with the Moeller-Trumbore division and no conversions; the rest."""
    loops = loops_of(ins)                                  # smallest first
    regs, used = [], []
    for tag in ("tri", "node"):
        for lo, hi in loops:
            nested = [(l2, h2) for l2, h2 in loops if (l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi]
            h, ops, un = hist_of(ins, lo, hi, exclude=nested + used)
            n = sum(h.values())
            cvt = sum(v for k, v in ops.items() if k.startswith("v_cvt_f32_ubyte"))
            div = any(k.startswith("v_div_") or k.startswith("v_rcp") for k in ops)
            if (tag == "node" and cvt >= 12) or (tag == "tri" and cvt == 0 and div and n >= 30):
                regs.append(region_report(f"{tag} loop [{lo:#x}, {hi:#x}]", h, ops, un, TRIPS[kind][tag])); used.append((lo, hi))
                break
    h, ops, un = hist_of(ins, exclude=used)
    regs.append(region_report("everything outside those loops (refill / ray set-up / leaf header / stack paging)", h, ops, un, TRIPS[kind]["other"]))
    return regs


def summarise(regs, straight=False):
    tot_c = tot_n = 0.0
    for r in regs:
        n = r["valu_instructions"]
        if not n:
            continue
        w = r["trips_weight"] if not straight else 1.0
        if r["region"].startswith("everything outside") and not straight:
            w = r["trips_weight"] * 0.1                    # that code is several alternative paths (refill OR page-out OR ...): a tenth of it per trip
        tot_c += w * n * r["cycles_per_instruction"]; tot_n += w * n
    cpis = [r["cycles_per_instruction"] for r in regs if r["valu_instructions"] >= 8]
    return {"cycles_per_instruction": tot_c / tot_n if tot_n else None, "range": [min(cpis), max(cpis)] if cpis else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "fluctus_amd", "libfluctus_hip.so"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "valu_roof.json"))
    args = ap.parse_args()
    from fluctus_amd import build
    tmp = tempfile.mkdtemp(prefix="valu_roof_")
    try:
        funcs = extract(args.lib, tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    want = {"k_trace4r<false, 0>": "trace", "k_trace4r<true, 1>": "shadow", "k_trace4r<true, 0>": "shadow",
            "k_shadow4<false, 1>": "shadow", "k_shadow4<false, 0>": "shadow", "k_extend4<false>": "trace",
            "k_logic<1, true, false>": "logic", "k_logic<31, true, false>": "logic", "k_logic<31, true, true>": "logic", "k_logic<0, false, false>": "logic",
            "k_logic<1, false, false>": "logic", "k_logic<31, false, false>": "logic",
            "k_raygen": "logic", "k_material_rest": "logic", "k_material<31>": "logic", "k_queue_scatter": "logic"}
    out = {"source_hash": build.source_hash(), "library": os.path.basename(args.lib), "model": {"pipe_time": COST, "rule": "cycles = max(S + T + P work, (all work) / 2); F and M issue on either pipe"},
           "ubench": ["profiles/r03_ubench_valu_rate.txt", "profiles/r03_ubench_valu_pairs.txt", "profiles/r03_ubench_valu_pairs_membership.txt"], "kernels": {}}
    for sym, f in funcs.items():
        name = demangle_short(sym)
        if name not in want:
            continue
        kind = want[name]
        if kind == "logic":
            h, ops, un = hist_of(f["ins"])
            regs = [region_report("whole kernel, every instruction once (straight-line, divergent: a wave runs every branch some lane takes)", h, ops, un, 1.0)]
            for lo, hi in loops_of(f["ins"]):
                h2, ops2, un2 = hist_of(f["ins"], lo, hi)
                if sum(h2.values()) >= 8:
                    regs.append(region_report(f"loop [{lo:#x}, {hi:#x}] (informative)", h2, ops2, un2, 0.0))
            s = summarise(regs[:1], straight=True)
            s["range"] = [min(r["cycles_per_instruction"] for r in regs), max(r["cycles_per_instruction"] for r in regs)]
        else:
            regs = traversal_kernel(f["ins"], kind)
            s = summarise(regs)
        out["kernels"][name] = dict(s, regions=regs)
    json.dump(out, open(args.out, "w"), indent=1)
    for k, v in sorted(out["kernels"].items()):
        print(f"{k:24s} cycles/instruction {v['cycles_per_instruction']:.2f}  range {v['range'][0]:.2f} .. {v['range'][1]:.2f}   " +
              " | ".join(f"{r['region'].split(' [')[0].split(',')[0][:28]}: {r['valu_instructions']} VALU @ {r['cycles_per_instruction']:.2f}" for r in v["regions"][:3]))


if __name__ == "__main__":
    main()
