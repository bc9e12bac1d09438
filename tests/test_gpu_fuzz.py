"""Fuzzing the call-sequence state machine of the boundary (csrc/api.hip).

The library defers and fuses behind the reference's entry points: flx_wf_logic is deferred until the next call shows whether the fused
logic + material pass can run, flx_wf_raygen is deferred along, the persistent extension kernel leaves RAW hit records that the next fused
pass -- or k_materialise, the moment anything else could look -- commits, the shadow kernel starts on a second stream when nothing but
genRays / materials / extension was enqueued since logic, the extension counter is bumped lazily, the block cursors are zeroed on demand.
The contract (single in-order queue, src/clcontext.cpp:765-895): whatever the host calls in whatever order, every observation equals what
the reference's separate kernels, run one by one in call order, would have produced.

The hand-picked call patterns of tests/test_gpu_wide.py::test_raw_hit_records_call_patterns cover ~20 sequences; here a seeded generator
draws >= 300 sequences of up to 12 calls over
    {logic(first), raygen, materials, extend, shadow, clear_queues, get_counters, finish, set_params, state_export, queue_read,
     set_option(fuse | refill_extend | extend_tree | fuse_set | overlap), pixel_index_update, end_iteration}
(kept inside what the reference itself defines: its queues hold NUM_TASKS entries -- logic once, genRays and the material kernels at most once
per clear -- and the microkernel integrator is not mixed in: on wavefront state it indexes materials[-1] in the reference's own kernels), runs each on
the device and on the oracle from a synchronised state, and compares the whole path state, all eight queues and the counters after every
sequence -- and at every observation inside one.  No RAW marker may ever be exported.  Coverage is reported as the set of
(phase before, call) pairs of the explicit state machine (flx_get_option "phase") that were exercised.
"""
import numpy as np
import pytest
import common
from common import COL, Q
from fluctus_amd import host, driver

pytestmark = pytest.mark.gpu

N_SEQ = 1200
MAX_LEN = 14


def _gen_sequence(rng):
    """A valid call sequence.  Validity = the reference's own limits: its queues hold NUM_TASKS entries, so between two clears `logic` runs
    at most once and genRays / the material kernels append their source queues at most once."""
    seq = []
    logic_done = raygen_done = mat_done = False
    length = int(rng.randint(3, MAX_LEN + 1))
    # bias toward the steady-state chain so that deferred / fused / RAW states are reached often, with random intruders
    chain = ["logic", "raygen", "materials", "extend", "shadow", "clear"]
    pos = 0
    while len(seq) < length:
        r = rng.rand()
        if r < 0.55:
            op = chain[pos % len(chain)]; pos += 1
        else:
            op = rng.choice(["logic", "raygen", "materials", "extend", "shadow", "clear", "counters", "finish", "params", "export", "qread",
                             "opt_fuse", "opt_refill", "opt_tree", "opt_fuseset", "opt_overlap", "pixidx", "end_iter", "pixels", "opt_shadow", "totals", "opt_regen", "opt_regroup", "opt_prep"])
        if op == "logic":
            if logic_done:
                continue
            logic_done = True
            seq.append(("logic", int(rng.rand() < 0.15)))
        elif op == "raygen":
            if raygen_done:
                continue
            raygen_done = True; seq.append(("raygen",))
        elif op == "materials":
            if mat_done:
                continue
            mat_done = True; seq.append(("materials",))
        elif op in ("clear", "end_iter"):
            logic_done = raygen_done = mat_done = False
            seq.append((op,))
        elif op == "params":
            seq.append(("params", int(rng.choice([2, 3, 5]))))
        elif op == "qread":
            seq.append(("qread", int(rng.randint(0, 8))))
        elif op == "opt_fuse":
            seq.append(("opt", "fuse", int(rng.randint(0, 2))))
        elif op == "opt_refill":
            seq.append(("opt", "refill_extend", int(rng.choice([0, 16 | (32 << 8), 8 | (16 << 8), 48]))))
        elif op == "opt_tree":
            seq.append(("opt", "extend_tree", int(rng.choice([2, 4]))))
        elif op == "opt_fuseset":
            seq.append(("opt", "fuse_set", int(rng.choice([1, 31]))))
        elif op == "opt_regen":                       # in-kernel regeneration of the fused RAW pass (logic.hip: REGEN) on / off
            seq.append(("opt", "regen", int(rng.randint(0, 2))))
        elif op == "opt_regroup":                     # all-types RAW pass with its material step sorted by BSDF type through LDS (logic.hip: LOGIC_REGROUP) on / off
            seq.append(("opt", "regroup", int(rng.randint(0, 2))))
        elif op == "opt_prep":                        # prepared regeneration: the seed-only half of genRays inside the fused RAW pass (logic.hip) on / off
            seq.append(("opt", "regen_prep", int(rng.randint(0, 2))))
        elif op == "opt_overlap":
            seq.append(("opt", "overlap", int(rng.choice([0, 1, 2]))))
        elif op == "pixidx":
            seq.append(("pixidx", int(rng.randint(0, 200))))
        elif op == "opt_shadow":
            if rng.rand() < 0.5:
                seq.append(("opt", "refill_shadow", int(rng.choice([0, 16 | (32 << 8)]))))
            else:
                seq.append(("opt", "shadow_tree", int(rng.choice([2, 4]))))
        else:
            seq.append((op,))
    return seq


# n: 4096 + 37 = a ragged last block; 17 x 256 = whole blocks, which the BSDF-regrouped all-types pass needs (block barriers: logic.hip LOGIC_REGROUP) -- round 6
@pytest.mark.parametrize("separate_queues,seed,n", [(1, 20260929, 4096 + 37), (0, 7, 4096 + 37), (1, 12345, 4096 + 37), (1, 606, 17 * 256), (0, 607, 17 * 256)])
def test_call_sequence_fuzz(separate_queues, seed, n):
    from fluctus_amd.device import HipContext
    from oracle.binding import OracleContext
    d = common.mixed_material_scene()
    w, h = 64, 48
    npix = w * h
    p = common.scene_params(d, w, h, maxBounces=5, useAreaLight=1, useEnvMap=1, wfSeparateQueues=separate_queues)
    env = host.synthetic_sky(64, 32)
    g, o = HipContext(n), OracleContext(n, threads=8)
    for c in (g, o):
        c.upload_scene(d); c.upload_envmap(env); c.set_params(p); driver.reset_renderer(c)
    g.set_option("ext_order", 0)                       # the separate kernels' extension-queue order: every queue comparable with ==
    cursor = 0                                         # the host-side pixel cursor, tracked here so that both contexts can be put on it
    g.set_option("extend_tree", 2)                     # (bit-exact closest hit for the warm-up: both framebuffers must hold the same samples)
    for _ in range(4):                                 # a populated steady state to start from
        cg, cnt = driver.benchmark_iteration(g, npix), driver.benchmark_iteration(o, npix)
        assert (cg == cnt).all()
        cursor = (cursor + int(cnt[Q.RAYGEN])) % npix
    assert common.fb_close(g.read_pixels(0), o.read_pixels(0))

    def set_cursor(c):
        c.pixel_index_reset(); c.pixel_index_update(npix, cursor)
    rng = np.random.RandomState(seed)
    covered = set()
    covered_order = set()                              # (ext_order, phase, call)
    exported_raw = 0
    nobs = 0

    def queue_equal(q, m, what):
        """Queue q against the oracle's.  The extension queue of a fused pass with ext_order 1 / 2 lists the same paths in path-id order instead of
        one segment per material queue (the reference's own order is whatever its atomic_inc produces): compared as a SET then, and -- the point
        of ext_order 2 -- every slot below the counter must hold a path (no gap left for a genRays that does not append: round 4's advisor)."""
        qa, qb = g.queue_read(q)[:m], o.queue_read(q)[:m]
        if q == Q.EXTENSION and g.get_option("ext_order") != 0:
            assert np.array_equal(np.sort(qa), np.sort(qb)), f"{what}: extension queue holds different paths"
        else:
            assert np.array_equal(qa, qb), f"{what}: queue {q} differs"

    def compare(what, queues=True):
        nonlocal exported_raw, nobs
        nobs += 1
        cg, co = g.get_counters(), o.get_counters(); g.finish()
        cg, co = np.array(cg, copy=True), np.array(co, copy=True)
        assert (cg == co).all(), f"{what}: counters {cg} vs {co}"
        if queues:
            for q in range(8):
                queue_equal(q, int(co[q]), what)
        sg = g.state_export()
        fails = common.state_diff(sg, o.state_export(), 0.0, 0.0)
        assert not fails, f"{what}: " + "; ".join(fails[:4])
        hi = sg.view(np.uint32)[COL.HIT_I]
        exported_raw += int((((hi >> 30) & 3) == 1).sum())

    def run(seq, fuse_set, ext_order, regen, regroup, prep, what, check_each=False):
        nonlocal cursor
        g.set_option("fuse", 1); g.set_option("extend_tree", 4); g.set_option("refill_extend", 16 | (32 << 8)); g.set_option("overlap", 2)
        g.set_option("shadow_tree", 4); g.set_option("refill_shadow", 0)
        g.set_option("fuse_set", fuse_set); g.set_option("ext_order", ext_order); g.set_option("regen", regen)
        g.set_option("regroup", regroup); g.set_option("regen_prep", prep)
        for c in (g, o):
            c.set_params(p)
        for k, op in enumerate(seq):
            ph = g.get_option("phase")
            covered.add((ph, op[0] if op[0] != "opt" else "opt:" + op[1]))
            covered_order.add((ext_order, ph & 7, op[0]))
            if op[0] == "logic":
                for c in (g, o): c.wf_logic(bool(op[1]))
            elif op[0] == "raygen":
                for c in (g, o): c.wf_raygen()
            elif op[0] == "materials":
                for c in (g, o): c.wf_materials()
            elif op[0] == "extend":
                for c in (g, o): c.wf_extend()
            elif op[0] == "shadow":
                for c in (g, o): c.wf_shadow()
            elif op[0] == "clear":
                for c in (g, o): c.clear_queues()
            elif op[0] == "end_iter":
                # the device-side end of an iteration (k_end_iteration: cursor advance by the raygen count + clear); the oracle does the same
                cnt = np.array(o.get_counters(), copy=True)
                g.end_iteration_async()
                o.clear_queues(); o.pixel_index_update(npix, int(cnt[Q.RAYGEN]))
                cursor = (cursor + int(cnt[Q.RAYGEN])) % npix
                set_cursor(g)                          # (k_end_iteration advanced the DEVICE copy; the host copy of the library follows here)
            elif op[0] == "counters":
                cg, co = g.get_counters(), o.get_counters(); g.finish()
                assert (np.array(cg) == np.array(co)).all(), f"{what} step {k}: counters {cg} vs {co}"
            elif op[0] == "finish":
                g.finish()
            elif op[0] == "params":
                p2 = p.copy(); p2["maxBounces"] = op[1]
                for c in (g, o): c.set_params(p2)
            elif op[0] == "export":
                compare(f"{what} step {k} (export)", queues=False)
            elif op[0] == "qread":
                queue_equal(op[1], int(np.array(o.get_counters())[op[1]]), f"{what} step {k}")
            elif op[0] == "pixels":
                if not check_each:                     # (a replay adds its samples a second time on both sides: still equal, but skip the noise)
                    assert common.fb_close(g.read_pixels(0), o.read_pixels(0)), f"{what} step {k}: framebuffers differ"
            elif op[0] == "totals":
                g.counter_totals(False)
            elif op[0] == "opt":
                g.set_option(op[1], op[2])
                if op[1] == "fuse_set":
                    g.set_option("ext_order", ext_order)          # (the sequence's order stays whatever the pass inlines)
            elif op[0] == "pixidx":
                for c in (g, o): c.pixel_index_update(npix, op[1])
                cursor = (cursor + op[1]) % npix
            if check_each:
                compare(f"{what}: REPLAY with a full comparison after every call -- first difference after call {k} {op}")
        compare(what)

    for s in range(N_SEQ):
        seq = _gen_sequence(rng)
        fuse_set = int(rng.choice([1, 31]))
        # the extension-queue order of the fused pass: 0 the separate kernels' segments | 1 continuing paths by id | 2 merged with the regenerated ones
        # (the shipped default of the diffuse-only pass: the scatter writes the regenerated paths' entries and the deferred genRays must not append)
        ext_order = int(rng.choice([0, 1, 2]))
        regen = int(rng.randint(0, 2))                 # the fused RAW pass regenerates its terminating paths itself (logic.hip: REGEN) | genRays does
        prep = int(rng.randint(0, 2))                  # prepared regeneration (the shipped default is 1)
        regroup = int(rng.randint(0, 2))               # the all-types RAW pass sorts its material step by BSDF type per block (takes effect with whole blocks of paths)
        # every sequence starts from the oracle's current state, queues cleared
        for c in (g, o):
            c.clear_queues()
        common.sync(g, o)
        start, cursor0 = o.state_export(), cursor
        what = f"sequence {s} (fuse_set {fuse_set}, ext_order {ext_order}, regen {regen}, regroup {regroup}, regen_prep {prep}) {seq}"
        try:
            run(seq, fuse_set, ext_order, regen, regroup, prep, what)
        except AssertionError:
            # locate the call: same sequence from the same state, everything compared after every call
            for c in (g, o):
                c.clear_queues(); c.state_import(start)
            cursor = cursor0
            for c in (g, o):
                set_cursor(c)
            run(seq, fuse_set, ext_order, regen, regroup, prep, what, check_each=True)
            raise
    assert exported_raw == 0, f"{exported_raw} RAW hit records were exported"
    phases = sorted({c[0] for c in covered})
    print(f"[fuzz] {N_SEQ} sequences, {nobs} full-state observations, {len(covered)} (phase, call) pairs over phases {phases}")
    # every phase of the state machine must have been entered, and every call class issued from at least three different phases
    assert len({ph & 7 for ph in phases}) == 6, phases          # all six call phases were entered
    by_call = {}
    for ph, call in covered:
        by_call.setdefault(call, set()).add(ph)
    for call in ("logic", "raygen", "materials", "extend", "shadow", "clear", "export", "counters", "params"):
        assert len(by_call.get(call, ())) >= 3, (call, by_call.get(call))
    # ... and the chain calls under every extension-queue order, each from at least two phases (ext_order 2 couples the fused scatter with the
    # deferred genRays: api.hip extOrderFor / runRaygen)
    for order in (0, 1, 2):
        for call in ("logic", "raygen", "materials", "extend", "clear", "end_iter"):
            phs = {ph for (eo, ph, c_) in covered_order if eo == order and c_ == call}
            assert len(phs) >= 2, (order, call, phs, sorted(covered_order))
    g.close()
