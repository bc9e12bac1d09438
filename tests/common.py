"""Shared helpers for the parity tests."""
import os
import numpy as np
from fluctus_amd import host, wire, driver
from fluctus_amd.wire import COL, Q, BXDF

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_ASSETS = "/root/reference/assets"

# integer-typed columns of the reference state layout (everything else is float)
INT_COLS = [COL.PHASE, COL.PATH_LEN, COL.SEED, COL.LAST_SPECULAR, COL.SHADOW_BLOCKED, COL.BACKFACE, COL.PIXEL_INDEX,
            COL.FIRST_DIFFUSE, COL.HIT_I, COL.AREA_LIGHT_HIT, COL.MAT_ID]
PAD_COLS = [3, 7, 11, 15, 19, 23, 27, 31, 35, 39, 43]      # float3 .w padding, never touched
COL_NAMES = {0: "orig", 4: "dir", 8: "shadowOrig", 12: "shadowDir", 16: "T", 20: "Ei", 24: "lastBsdf", 28: "lastEmission",
             32: "lastT", 36: "P", 40: "N", 44: "uv", 46: "phase", 47: "lastPdfW", 48: "pathLen", 49: "seed",
             50: "lastSpecular", 51: "shadowRayBlocked", 52: "backfaceHit", 53: "pixelIndex", 54: "firstDiffuseHit",
             55: "lastPdfDirect", 56: "lastPdfImplicit", 57: "lastCosTh", 58: "lastLightPickProb", 59: "shadowRayLen",
             60: "t", 61: "i", 62: "areaLightHit", 63: "matId"}


def colname(c):
    base = max(k for k in COL_NAMES if k <= c)
    return COL_NAMES[base] + (f"[{c - base}]" if c != base else "")


def small_mesh_scene(n=6, mats=None, seed=3):
    """A few hundred triangles: ground grid + a bumpy sphere above it, built in numpy (wire format)."""
    rng = np.random.RandomState(seed)
    tris = []

    def tri(p0, p1, p2, n0, n1, n2, t0, t1, t2, m):
        t = np.zeros((), wire.TRIANGLE)
        for v, p, nn, tt in (("v0", p0, n0, t0), ("v1", p1, n1, t1), ("v2", p2, n2, t2)):
            t[v]["p"]["x"], t[v]["p"]["y"], t[v]["p"]["z"] = p
            t[v]["n"]["x"], t[v]["n"]["y"], t[v]["n"]["z"] = nn
            t[v]["t"]["x"], t[v]["t"]["y"] = tt
        t["matId"] = m
        tris.append(t)

    g = 2 * n
    for j in range(g):
        for i in range(g):
            x0, x1 = -2 + 4 * i / g, -2 + 4 * (i + 1) / g
            z0, z1 = -2 + 4 * j / g, -2 + 4 * (j + 1) / g
            up = (0, 1, 0)
            m = 1 + (i + j) % 2 if mats is not None and mats > 2 else 0
            tri((x0, 0, z0), (x1, 0, z0), (x1, 0, z1), up, up, up, (i / g, j / g), ((i + 1) / g, j / g), ((i + 1) / g, (j + 1) / g), m)
            tri((x0, 0, z0), (x1, 0, z1), (x0, 0, z1), up, up, up, (i / g, j / g), ((i + 1) / g, (j + 1) / g), (i / g, (j + 1) / g), m)

    def sph(u, v, c, r):
        th, ph = u * 2 * np.pi, v * np.pi
        d = np.array([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)])
        return c + r * d, d

    nobj = 1 if mats is None else max(1, mats - 3)
    for k in range(nobj):
        c = np.array([-1.2 + 2.4 * (k + 0.5) / nobj if nobj > 1 else 0.0, 0.8, 0.0 + 0.3 * ((k % 2) * 2 - 1) * (nobj > 1)])
        r = 0.7 if nobj == 1 else 0.32
        m = 0 if mats is None else 3 + k
        for j in range(n):
            for i in range(2 * n):
                a, na = sph(i / (2 * n), j / n, c, r)
                b, nb = sph((i + 1) / (2 * n), j / n, c, r)
                cc, nc = sph((i + 1) / (2 * n), (j + 1) / n, c, r)
                dd, nd = sph(i / (2 * n), (j + 1) / n, c, r)
                ta, tb, tc, td = (i / (2 * n), j / n), ((i + 1) / (2 * n), j / n), ((i + 1) / (2 * n), (j + 1) / n), (i / (2 * n), (j + 1) / n)
                if j > 0:
                    tri(a, b, cc, na, nb, nc, ta, tb, tc, m)
                if j < n - 1:
                    tri(a, cc, dd, na, nc, nd, ta, tc, td, m)
    d = host.SceneData()
    d.tris = np.array(tris, wire.TRIANGLE)
    return d


def make_material(type_, kd=(0.6, 0.5, 0.4), ks=(0.5, 0.5, 0.5), ns=100.0, ni=1.5, map_kd=-1, map_ks=-1, map_n=-1):
    m = np.zeros((), wire.MATERIAL)
    m["Kd"]["x"], m["Kd"]["y"], m["Kd"]["z"] = kd
    m["Ks"]["x"], m["Ks"]["y"], m["Ks"]["z"] = ks
    m["Ns"], m["Ni"], m["map_Kd"], m["map_Ks"], m["map_N"], m["type"] = ns, ni, map_kd, map_ks, map_n, type_
    return m


def default_material():
    return make_material(BXDF.DIFFUSE, kd=(0.64, 0.64, 0.64), ks=(0, 0, 0), ns=700.0, ni=1.8)


def checker_texture(size=16, seed=1):
    rng = np.random.RandomState(seed)
    t = rng.randint(0, 256, size=(size, size, 4)).astype(np.uint8)
    t[..., 3] = 255
    return t


def mixed_material_scene(with_textures=True):
    """Small scene exercising all six BSDF types (+ textures and a normal map) -- the material-queue stress case."""
    d = small_mesh_scene(n=6, mats=9)
    mats = [default_material(),
            make_material(BXDF.DIFFUSE, kd=(0.7, 0.3, 0.2), map_kd=0 if with_textures else -1),
            make_material(BXDF.GLOSSY, kd=(0.2, 0.5, 0.7), ks=(0, 0, 0), ns=300.0, ni=1.5, map_n=1 if with_textures else -1),
            make_material(BXDF.DIFFUSE, kd=(0.5, 0.5, 0.5)),
            make_material(BXDF.GLOSSY, kd=(0.6, 0.2, 0.2), ks=(0.4, 0.4, 0.4), ns=80.0, ni=0.0),
            make_material(BXDF.GGX_ROUGH_REFLECTION, ks=(0.9, 0.8, 0.5), ns=60.0, ni=1.0),
            make_material(BXDF.IDEAL_REFLECTION, ks=(0.9, 0.9, 0.9)),
            make_material(BXDF.GGX_ROUGH_DIELECTRIC, ks=(0.95, 0.95, 0.95), ns=400.0, ni=1.5),
            make_material(BXDF.IDEAL_DIELECTRIC, ks=(0.9, 0.95, 1.0), ni=1.5)]
    d.materials = np.array(mats, wire.MATERIAL)
    if with_textures:
        t0, t1 = checker_texture(16, 1), checker_texture(8, 2)
        d.texdesc = np.zeros(2, wire.TEXDESC)
        d.texdesc[0] = (0, 16, 16)
        d.texdesc[1] = (t0.size, 8, 8)
        d.texdata = np.concatenate([t0.reshape(-1), t1.reshape(-1)]).astype(np.uint8)
    else:
        d.texdesc = np.zeros(0, wire.TEXDESC)
        d.texdata = np.zeros(0, np.uint8)
    host.build_bvh(d, "sbvh")
    return d


def simple_scene():
    d = small_mesh_scene(n=6)
    d.materials = np.array([default_material()], wire.MATERIAL)
    d.texdesc = np.zeros(0, wire.TEXDESC)
    d.texdata = np.zeros(0, np.uint8)
    host.build_bvh(d, "sbvh")
    return d


def scene_params(d, w, h, **kw):
    p = wire.default_params(w, h, d.world_radius, d.tris.size)
    wire.look_at(p, (0.0, 1.6, 3.2), (0.0, 0.5, 0.0))
    al = p["areaLight"]
    al["pos"]["x"], al["pos"]["y"], al["pos"]["z"] = 1.6, 1.8, 0.4
    for k, v in kw.items():
        p[k] = v
    return p


def fixture_scene(z):
    """SceneData of a golden fixture: its own arrays, or -- `scene_file` -- the triangle / material / texture arrays of another fixture
    (tests/golden/egyptcat_scene.npz) with the fixture's own BVH."""
    s = np.load(os.path.join(GOLDEN, str(z["scene_file"]))) if "scene_file" in z.files else z
    d = host.SceneData()
    d.tris = s["tris"].view(wire.TRIANGLE).reshape(-1)
    d.materials = s["materials"].view(wire.MATERIAL).reshape(-1)
    d.texdesc = s["texdesc"].view(wire.TEXDESC).reshape(-1) if s["texdesc"].size else np.zeros(0, wire.TEXDESC)
    d.texdata = s["texdata"]
    if "nodes" in z.files:
        d.nodes = z["nodes"].view(wire.NODE).reshape(-1)
        d.indices = z["indices"]
    return d


def egyptcat_scene():
    """tests/golden/egyptcat_scene.npz (scripts/make_egyptcat_fixture.py): the reference's assets/egyptcat/egyptcat.obj as loaded by
    host/scene.cpp, with the SBVH built here."""
    d = fixture_scene(np.load(os.path.join(GOLDEN, "egyptcat_scene.npz")))
    host.build_bvh(d, "sbvh")
    return d


def sync(dst, src):
    """Make dst's path state / queues / counters identical to src's."""
    dst.state_import(src.state_export())
    cnt = src.get_counters()
    if hasattr(src, "finish"):
        src.finish()
    cnt = np.array(cnt, copy=True)
    for q in range(Q.NUM):                          # all eight (until round 5 this loop stopped before the delta queue)
        dst.queue_write(q, src.queue_read(q))
    dst.set_counters(cnt)


def sharp_lobe_paths(d, state):
    """Paths whose hit material is a glossy / GGX lobe with Ns >= 1e4 (alpha = sqrt(2 / (2 + Ns)) <= 0.014; egyptcat.mtl: Ns 100000).
    The GGX density D = a^2 / (pi ((n.h)^2 (a^2 - 1) + 1)^2) is evaluated at n.h = 1 - O(a^2): one ulp of cos(theta) (6e-8) against
    a^2 = 2e-5 moves the denominator by ~1 %, so the REFERENCE's libm build and the oracle's flx_math can differ by that much in the pdf of a
    sampled direction there (observed: 0.5 % on one path of 4 096) -- the quantity is ill-conditioned in fp32, in the reference itself.
    Device vs oracle stays bit-exact (same flx_math)."""
    mid = np.clip(state.view(np.int32)[COL.MAT_ID], 0, d.materials.size - 1)
    m = d.materials[mid]
    return (m["Ns"] >= 1.0e4) & np.isin(m["type"], (BXDF.GLOSSY, BXDF.GGX_ROUGH_REFLECTION, BXDF.GGX_ROUGH_DIELECTRIC))


def state_diff(sa, sb, rtol, atol, skip_cols=(), mask=None, col_rtol=None):
    """Compare two (64, N) reference-layout states. Integer columns exact, float columns within tol (col_rtol: {column: per-path
    rtol array or scalar} overrides).  Returns a list of human-readable failures."""
    fails = []
    ia, ib = sa.view(np.uint32), sb.view(np.uint32)
    for c in range(64):
        if c in PAD_COLS or c in skip_cols or c == COL.PHASE:
            continue
        a, b = (ia[c], ib[c]) if c in INT_COLS else (sa[c], sb[c])
        if mask is not None:
            a, b = a[mask], b[mask]
        if c in INT_COLS:
            bad = a != b
        else:
            rt = rtol
            if col_rtol is not None and c in col_rtol:
                rt = np.asarray(col_rtol[c])
                if rt.ndim and mask is not None:
                    rt = rt[mask]
            with np.errstate(all="ignore"):
                bad = ~(np.abs(a - b) <= atol + rt * np.abs(b))
            bad &= ~((a == b) | (np.isnan(a) & np.isnan(b)))
        if bad.any():
            j = int(np.argmax(bad))
            fails.append(f"col {c} ({colname(c)}): {int(bad.sum())} mismatches, first at {j}: {a[j]!r} vs {b[j]!r}")
    return fails


def fb_close(a, b):
    """Raw accumulation buffers (rgb sum, sample count): counts exact; a sum of N non-negative fp32 terms added in ANY order (float
    atomics on the device, ascending path id in the oracle) differs by at most (N - 1) * 2^-24 relative -- the bound, doubled, with N
    the pixel's own sample count.  (A fixed rtol 1e-6 held for round 1's tests but is only a typical value: ~sqrt(N) * 6e-8.)"""
    a, b = np.asarray(a), np.asarray(b)
    if not np.array_equal(a[:, 3], b[:, 3]):
        return False
    n = np.maximum(b[:, 3:4], 1.0)
    return bool((np.abs(a[:, :3] - b[:, :3]) <= 2.0 * n * 2.0 ** -24 * np.abs(b[:, :3]) + 1e-7).all())
