"""Host-side preparation (C++ libfluctus_host.so): loaders, BVH/SBVH builders, env-map tables, arithmetic contract."""
import ctypes as C
import os
import numpy as np
import pytest
import common
from fluctus_amd import host, wire
from oracle import binding as ob

REF = "/root/reference/assets"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_ref_assets = pytest.mark.skipif(not os.path.isdir(REF), reason="reference assets only exist in the build container")


def _check_bvh(d):
    nodes, idx = d.nodes, d.indices
    n = nodes.size
    seen = np.zeros(d.tris.size, bool)
    leaf_slots = 0
    stack = [0]
    visited = 0
    tmin = np.minimum(np.minimum(_p(d, "v0"), _p(d, "v1")), _p(d, "v2"))
    tmax = np.maximum(np.maximum(_p(d, "v0"), _p(d, "v1")), _p(d, "v2"))
    while stack:
        i = stack.pop()
        visited += 1
        nd = nodes[i]
        bmin = np.array([nd["bmin"]["x"], nd["bmin"]["y"], nd["bmin"]["z"]])
        bmax = np.array([nd["bmax"]["x"], nd["bmax"]["y"], nd["bmax"]["z"]])
        if nd["nPrims"]:
            s, c = int(nd["iStartOrRight"]), int(nd["nPrims"])
            assert s + c <= idx.size
            leaf_slots += c
            for t in idx[s:s + c]:
                seen[t] = True
                # spatial splits clip references, so a leaf box only has to OVERLAP its triangles' boxes
                assert (tmin[t] <= bmax + 1e-5).all() and (tmax[t] >= bmin - 1e-5).all()
        else:
            l, r = i + 1, int(nd["iStartOrRight"])
            assert l < n and r < n and nodes[l]["parent"] == i and nodes[r]["parent"] == i
            for ch in (l, r):
                cm = np.array([nodes[ch]["bmin"]["x"], nodes[ch]["bmin"]["y"], nodes[ch]["bmin"]["z"]])
                cM = np.array([nodes[ch]["bmax"]["x"], nodes[ch]["bmax"]["y"], nodes[ch]["bmax"]["z"]])
                assert (cm >= bmin - 1e-5).all() and (cM <= bmax + 1e-5).all()
            stack += [r, l]
    assert visited == n and seen.all() and leaf_slots == idx.size


def _p(d, v):
    return np.stack([d.tris[v]["p"]["x"], d.tris[v]["p"]["y"], d.tris[v]["p"]["z"]], 1)


@pytest.mark.parametrize("mode", ["sbvh", "sah", "binned"])
def test_bvh_builders_produce_valid_trees(mode):
    d = common.small_mesh_scene(n=8)
    host.build_bvh(d, mode)
    _check_bvh(d)
    assert d.nodes[0]["parent"] == -1
    assert d.bvh_metrics["depth"] <= 64
    ext = np.array([d.nodes[0]["bmax"][k] - d.nodes[0]["bmin"][k] for k in "xyz"])
    assert abs(d.world_radius - 0.5 * np.linalg.norm(ext)) < 1e-5      # reference: src/tracer.cpp:66-67


@pytest.mark.parametrize("gen,tris,seed", [("kitchen", 12000, 42), ("conference", 8000, 43), ("courtyard", 16000, 44)])
def test_sbvh_parallel_build_is_the_serial_tree(gen, tris, seed):
    """The parallel SBVH build (top of the tree: one node at a time with parallel sort + binning; below: independent subtrees as
    parallel jobs) must return the serial recursion's node and index arrays byte for byte, for any thread count and job size --
    the reference's builder (src/sbvh.cpp:105-157) is serial, and the tree decides traversal order and exact-tie winners."""
    d = host.generate_scene(gen, tris, seed)
    host.build_bvh(d, "sbvh", threads=1)
    nodes, indices, met = d.nodes.copy(), d.indices.copy(), dict(d.bvh_metrics)
    assert met["spatial_splits"] > 0
    for threads, job in ((0, 0), (3, 500), (8, 64)):
        host.build_bvh(d, "sbvh", threads=threads, job_size=job)
        assert np.array_equal(d.nodes.view(np.uint8), nodes.view(np.uint8)), (threads, job)
        assert np.array_equal(d.indices, indices), (threads, job)
        assert d.bvh_metrics == met, (threads, job)


@pytest.mark.parametrize("gen,tris,seed,mode", [("kitchen", 30000, 42, "sbvh"), ("conference", 20000, 43, "sbvh"),
                                                ("courtyard", 40000, 44, "binned"), ("kitchen", 3000, 7, "sah")])
def test_wide_tree_invariants(gen, tris, seed, mode):
    """The kernels' 4-wide quantised tree (csrc/flx_wide.h, built at flx_upload_scene from the reference-format binary tree) on the
    CPU: conservative boxes in real arithmetic, each binary leaf exactly once with the leaf's exact box / count / triangle order (what
    makes any-hit results identical and closest hits differ on ties only), each wide node once, and the SAH collapse fills the slots."""
    d = host.generate_scene(gen, tris, seed)
    host.build_bvh(d, mode)
    info = host.wide_tree_check(d)
    nleaf = int((d.nodes["nPrims"] > 0).sum())
    assert info["leaves"] == nleaf
    assert info["leaf_float4s"] == 5 + 2 * nleaf + 3 * d.indices.size
    # k-ary tree with L leaves: (L - 1) / 3 <= wide nodes <= L - 1; the collapse should get close to the lower end
    assert (nleaf - 1 + 2) // 3 <= info["wide_nodes"] <= nleaf - 1
    assert info["wide_nodes"] <= 0.45 * nleaf
    assert sum(info["slots_used"].values()) == info["wide_nodes"]
    assert sum((k - 1) * v for k, v in info["slots_used"].items()) == nleaf - 1
    depth = d.bvh_metrics["depth"]
    assert 1 <= info["max_stack"] <= 3 * depth + 1


def test_wide_tree_degenerate_inputs():
    """One-leaf scene, a two-leaf tree, a maximally unbalanced chain, and malformed arrays (must be refused, not crash)."""
    from fluctus_amd.wire import NODE
    d = common.small_mesh_scene(n=2)
    d.tris = d.tris[:3].copy()
    host.build_bvh(d, "sah")
    assert d.nodes.size == 1
    info = host.wide_tree_check(d)
    assert info["leaves"] == 1 and info["max_stack"] == 1
    # chain: leaf i holds triangle i; inner nodes 0, 2, 4 ... each have {leaf, rest}
    d = common.small_mesh_scene(n=4)
    nt = 40
    d.tris = d.tris[:nt].copy()
    P = np.stack([np.stack([d.tris[v]["p"][a] for a in "xyz"], -1) for v in ("v0", "v1", "v2")], 1)       # (nt, 3 verts, 3)
    tmin, tmax = P.min(1), P.max(1)
    nodes = np.zeros(2 * nt - 1, NODE)
    for k in range(nt - 1):                                      # inner node 2k: left leaf 2k+1 (tri k), right 2k+2
        i = 2 * k
        lo, hi = tmin[k:].min(0), tmax[k:].max(0)
        for a, ax in enumerate("xyz"):
            nodes[i]["bmin"][ax], nodes[i]["bmax"][ax] = lo[a], hi[a]
            nodes[i + 1]["bmin"][ax], nodes[i + 1]["bmax"][ax] = tmin[k][a], tmax[k][a]
        nodes[i]["iStartOrRight"] = i + 2
        nodes[i + 1]["iStartOrRight"], nodes[i + 1]["nPrims"] = k, 1
    last = 2 * nt - 2
    for a, ax in enumerate("xyz"):
        nodes[last]["bmin"][ax], nodes[last]["bmax"][ax] = tmin[nt - 1][a], tmax[nt - 1][a]
    nodes[last]["iStartOrRight"], nodes[last]["nPrims"] = nt - 1, 1
    d.nodes, d.indices = nodes, np.arange(nt, dtype=np.uint32)
    info = host.wide_tree_check(d)
    assert info["leaves"] == nt and info["wide_nodes"] == (nt - 1 + 2) // 3
    # the nearest child can be the inner one at every level, leaving three leaves pending per wide node: the bound the spill buffer
    # is sized from must cover that
    assert info["max_stack"] == 3 * info["wide_nodes"] + 1
    # malformed: right child pointing backwards, leaf range outside the index list, triangle index out of range
    bad = nodes.copy(); bad[0]["iStartOrRight"] = 0
    d.nodes = bad
    with pytest.raises(Exception, match="child index"):
        host.wide_tree_check(d)
    d.nodes = nodes; d.indices = np.arange(nt - 1, dtype=np.uint32)
    with pytest.raises(Exception, match="index list"):
        host.wide_tree_check(d)
    d.indices = np.arange(nt, dtype=np.uint32) + 1
    with pytest.raises(Exception, match="triangle index"):
        host.wide_tree_check(d)
    # non-finite input (round 2's advisor): an infinite box passes `min <= max`, its extent is inf and the grid exponent would come out
    # of log2(inf) converted to int; a NaN / inf vertex would be copied into a leaf block.  Both are refused with a message of their own,
    # and so is a box beyond +-2^62, where (o - orig) / dir of the node test could overflow (flx_trace4.h: WRay::setup)
    d.indices = np.arange(nt, dtype=np.uint32)
    for val, where in ((np.inf, "bmax"), (-np.inf, "bmin"), (1e19, "bmax")):
        bad = nodes.copy(); bad[4][where]["y"] = val; bad[0][where]["y"] = val; bad[2][where]["y"] = val
        d.nodes = bad
        with pytest.raises(Exception, match="not finite or beyond"):
            host.wide_tree_check(d)
    d.nodes = nodes
    keep = d.tris.copy()
    for val in (np.nan, np.inf):
        d.tris = keep.copy(); d.tris[7]["v1"]["p"]["z"] = val
        with pytest.raises(Exception, match="NaN or infinite vertex"):
            host.wide_tree_check(d)
    d.tris = keep
    assert host.wide_tree_check(d)["leaves"] == nt


def _soup(n, seed, long_share=0.3):
    """n random triangles in general position (no axis-aligned or coplanar sets: the reference's binning divides by the node extent),
    a share of them long and thin so that spatial splits pay."""
    rng = np.random.RandomState(seed)
    c = rng.uniform(-1, 1, (n, 1, 3))
    size = np.where(rng.rand(n, 1, 1) < long_share, rng.uniform(0.8, 2.0, (n, 1, 1)), rng.uniform(0.03, 0.25, (n, 1, 1)))
    v = (c + rng.normal(size=(n, 3, 3)) * size * np.array([1.0, 0.15, 0.4])).astype(np.float32)
    d = host.SceneData()
    t = np.zeros(n, wire.TRIANGLE)
    for k, name in enumerate(("v0", "v1", "v2")):
        for a, ax in enumerate("xyz"):
            t[name]["p"][ax] = v[:, k, a]
        t[name]["n"]["y"] = 1.0
    d.tris = t
    return d, v


@pytest.mark.parametrize("n,seed", [(60, 1), (140, 2), (220, 3)])
def test_sbvh_matches_an_independent_restatement(n, seed):
    """host/bvh.cpp's Mode::SBVH against tests/sbvh_restatement.py -- the reference's builder (src/sbvh.cpp) restated a second time,
    in Python over fp32 scalars -- node for node and index for index (boxes bit for bit), on triangle soups where spatial splits,
    unsplitting and duplication all occur.  The reference's own sbvh.cpp cannot be built here (progressview.hpp -> nanogui)."""
    from sbvh_restatement import SBVH
    d, verts = _soup(n, seed)
    host.build_bvh(d, "sbvh", threads=1)
    ref = SBVH(verts)
    assert ref.spatial > 0 and ref.duplicates > 0, "the scene is meant to exercise spatial splits"
    assert d.bvh_metrics["duplicates"] == ref.duplicates and d.bvh_metrics["splits"] == ref.splits and d.bvh_metrics["depth"] == ref.depth
    assert np.array_equal(d.indices, np.array(ref.indices, np.uint32))
    assert d.nodes.size == len(ref.nodes)
    for i, (box, parent, link, nprims) in enumerate(ref.nodes):
        nd = d.nodes[i]
        got = np.array([nd["bmin"]["x"], nd["bmin"]["y"], nd["bmin"]["z"], nd["bmax"]["x"], nd["bmax"]["y"], nd["bmax"]["z"]], np.float32)
        want = np.array(box.mn + box.mx, np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"node {i}: box"
        assert int(nd["parent"]) == parent and int(nd["iStartOrRight"]) == link and int(nd["nPrims"]) == nprims, f"node {i}"


def _sbvh_digest_of_host_builder(d):
    sys_path = os.path.join(ROOT, "scripts")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_sbvh_golden", os.path.join(sys_path, "make_sbvh_golden.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_sbvh_matches_the_restatement_on_teapot():
    """Round 2's verdict (weak #10): the same comparison on a REAL mesh -- BASELINE configs[0]'s teapot.ply, 3 206 triangles, rebuilt live
    by the Python restatement (~1.5 min): node array, index list, split / duplicate counts identical."""
    if not os.path.exists(REF + "/teapot.ply"):
        pytest.skip("teapot.ply not in the checkout")
    m = _sbvh_digest_of_host_builder(None)
    d = host.load_scene(REF + "/teapot.ply")
    ref, nodes, idx = m.restatement_arrays(d)
    host.build_bvh(d, "sbvh", threads=1)
    assert d.bvh_metrics["duplicates"] == ref.duplicates and d.bvh_metrics["splits"] == ref.splits and d.bvh_metrics["depth"] == ref.depth
    assert np.array_equal(d.indices, idx) and d.nodes.size == nodes.size
    for f in ("parent", "iStartOrRight", "nPrims"):
        assert np.array_equal(d.nodes[f], nodes[f]), f
    for b in ("bmin", "bmax"):
        for a in "xyz":
            assert np.array_equal(d.nodes[b][a].view(np.uint32), nodes[b][a].view(np.uint32)), (b, a)


def test_sbvh_matches_the_restatement_digest_on_conference_38k():
    """... and on the 38 k-triangle conference-proc mesh the SAH pin uses.  The Python restatement needs ~half an hour for it, so its output is
    pinned as a digest (scripts/make_sbvh_golden.py -> tests/golden/sbvh_restatement_digest.json: SHA-256 of boxes, links, index list +
    split / duplicate counts); host/bvh.cpp's serial AND parallel builds must hash to the same values."""
    import json
    path = os.path.join(ROOT, "tests", "golden", "sbvh_restatement_digest.json")
    if not os.path.exists(path):
        pytest.skip("digest fixture not generated")
    want = json.load(open(path))["conference-38k"]
    m = _sbvh_digest_of_host_builder(None)
    for threads in (1, 8):
        d = host.generate_scene("conference", 6000, 43)
        assert d.tris.size == want["triangles"]
        host.build_bvh(d, "sbvh", threads=threads)
        got = m.digest(d.nodes, d.indices)
        for k in ("nodes", "indices", "boxes_sha256", "links_sha256", "indices_sha256"):
            assert got[k] == want[k], (threads, k, got[k], want[k])
        assert d.bvh_metrics["duplicates"] == want["duplicates"] and d.bvh_metrics["splits"] == want["splits"] and d.bvh_metrics["depth"] == want["depth"]


def test_sbvh_creates_duplicates_only_with_spatial_splits():
    d = host.generate_scene("conference", 20000, 43)
    host.build_bvh(d, "sbvh")
    assert d.indices.size == d.tris.size + d.bvh_metrics["duplicates"]
    _check_bvh(d)


def test_builders_agree_on_traversal_results():
    """Closest hits do not depend on the tree: oracle traversal over SBVH vs SAH vs binned trees."""
    from fluctus_amd import driver
    res = []
    for mode in ("sbvh", "sah", "binned"):
        d = common.small_mesh_scene(n=8)
        d.materials = np.array([common.default_material()], wire.MATERIAL)
        d.texdesc = np.zeros(0, wire.TEXDESC); d.texdata = np.zeros(0, np.uint8)
        host.build_bvh(d, mode)
        p = common.scene_params(d, 64, 48)
        p["worldRadius"] = 3.0
        c = ob.OracleContext(64 * 48)
        c.upload_scene(d); c.set_params(p)
        c.pixel_index_reset(); c.wf_reset(); c.wf_raygen(); c.wf_extend()
        st = c.state_export()
        res.append((st[common.COL.HIT_T].copy(), st.view(np.uint32)[common.COL.HIT_I].copy()))
    for t, i in res[1:]:
        assert np.array_equal(t, res[0][0]) and np.array_equal(i, res[0][1])


@needs_ref_assets
def test_ply_loader_teapot():
    d = host.load_scene(REF + "/teapot.ply")
    assert d.tris.size == 3206 and d.materials.size == 1
    m = d.materials[0]
    assert m["type"] == wire.BXDF.DIFFUSE and abs(m["Kd"]["x"] - 0.64) < 1e-7 and m["Ni"] == np.float32(1.8) and m["Ns"] == 700.0
    n = np.stack([d.tris["v0"]["n"]["x"], d.tris["v0"]["n"]["y"], d.tris["v0"]["n"]["z"]], 1)
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-3)  # per-vertex normals were read
    assert (d.tris["matId"] == 0).all()


def _ply_by_the_reference_rules(path):
    """The reference's loadPlyModel restated in numpy (src/scene.cpp:421-552): header = element / property NAMES (types ignored), one
    line per vertex, every token atof'd and keyed by its property name (a name listed twice keeps the LAST value), faces '3 a b c' or
    '4 a b c d' -> (a, b, c), (c, d, a); unpackIndexedData (:814-860): normals by vertex index, or the face normal when the file has none."""
    lines = open(path).read().split("\n")
    elements, i = [], 0
    while True:
        tok = lines[i].split(); i += 1
        if tok and tok[0] == "element":
            elements.append([tok[1], int(tok[2]), []])
        elif tok and tok[0] == "property":
            elements[-1][2].append(tok[2] if tok[1] != "list" else tok[-1])      # iss >> type >> name: for a list property `name` is the count type
        elif tok and tok[0] == "end_header":
            break
    pos = nrm = None
    faces = []
    for name, n, props in elements:
        if name == "vertex":
            rows = np.array([[float(t) for t in lines[i + k].split()[:len(props)]] for k in range(n)], np.float64).astype(np.float32)
            col = {p_: j for j, p_ in enumerate(props)}                            # last occurrence wins, like std::map assignment
            pos = rows[:, [col["x"], col["y"], col["z"]]]
            nrm = rows[:, [col["nx"], col["ny"], col["nz"]]] if "nx" in col else None
        elif name == "face":
            for k in range(n):
                t = [int(x) for x in lines[i + k].split()]
                if t[0] == 3:
                    faces.append((t[1], t[2], t[3]))
                elif t[0] == 4:
                    faces.append((t[1], t[2], t[3])); faces.append((t[3], t[4], t[1]))
        i += n
    f = np.array(faces)
    P = pos[f]                                                                       # (ntri, 3, 3)
    return P, (nrm[f] if nrm is not None else None)                                 # no normals: the face normal (checked on a known quad below)


@needs_ref_assets
def test_ply_loader_is_the_reference_loaders_mesh(tmp_path):
    """Every triangle of teapot.ply -- vertex order, positions and per-vertex normals bit for bit -- against the reference loader's
    rules restated independently in numpy (the reference's scene.cpp cannot be built here: pbrtParser / nanogui headers).  The file
    exercises the odd parts: 16 float properties per vertex of which several NAMES repeat, and an integer list for the faces."""
    path = REF + "/teapot.ply"
    d = host.load_scene(path)
    P, N = _ply_by_the_reference_rules(path)
    assert d.tris.size == P.shape[0] == 3206
    for k, v in enumerate(("v0", "v1", "v2")):
        got_p = np.stack([d.tris[v]["p"][a] for a in "xyz"], 1)
        got_n = np.stack([d.tris[v]["n"][a] for a in "xyz"], 1)
        assert np.array_equal(got_p.view(np.uint32), P[:, k].view(np.uint32)), v
        assert np.array_equal(got_n.view(np.uint32), N[:, k].view(np.uint32)), v
    # quads and files without normals (not in the asset): '4 a b c d' -> (a, b, c), (c, d, a); flat normals
    q = tmp_path / "quad.ply"
    q.write_text("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    dq = host.load_scene(str(q))
    Pq, _ = _ply_by_the_reference_rules(str(q))
    assert dq.tris.size == 2
    for k, v in enumerate(("v0", "v1", "v2")):
        assert np.array_equal(np.stack([dq.tris[v]["p"][a] for a in "xyz"], 1), Pq[:, k])
        assert np.array_equal(np.stack([dq.tris[v]["n"][a] for a in "xyz"], 1), np.array([[0, 0, 1], [0, 0, 1]], np.float32))


@needs_ref_assets
def test_obj_mtl_loader_egyptcat():
    d = host.load_scene(REF + "/egyptcat/egyptcat.obj")
    faces = [l.split()[1:] for l in open(REF + "/egyptcat/egyptcat.obj") if l.startswith("f ")]
    assert len(faces) == 16026                                     # 16 026 `f` lines
    assert d.tris.size == sum(len(f) - 2 for f in faces)           # polygons are fan-triangulated (tinyobj triangulate=true)
    assert d.materials.size == 4                                   # default + 3 from egyptcat.mtl
    assert d.materials[1]["type"] == wire.BXDF.GLOSSY and d.materials[1]["Ns"] == 100000.0   # `shader glossy`
    assert d.materials[2]["type"] == wire.BXDF.DIFFUSE
    assert set(np.unique(d.tris["matId"])) <= {0, 1, 2, 3}


def test_obj_loader_edge_cases(tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 0 0\nshader rough_reflection\nNs 50\nnewmtl b\nKd 0 1 0\nNi 1.5\nshader ideal_dielectric\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\n"
                                    "f 1 2 3\nusemtl a\nf 1/1/1 2/2/1 3/3/1 4/1/1\nusemtl b\nf -4//1 -3//1 -2//1\nusemtl missing\nf 1 2 4\n")
    d = host.load_scene(str(tmp_path / "m.obj"))
    assert d.tris.size == 5                                        # 1 + quad fan (2) + 1 + 1
    assert list(d.tris["matId"]) == [0, 1, 1, 2, 0]                # -1 -> default material 0, mtl index + 1 otherwise
    assert d.materials[1]["type"] == wire.BXDF.GGX_ROUGH_REFLECTION and d.materials[2]["type"] == wire.BXDF.IDEAL_DIELECTRIC
    assert d.tris[0]["v0"]["n"]["z"] == 1.0                        # flat normal generated when normals are missing
    assert d.tris[1]["v1"]["t"]["x"] == 1.0                        # texcoords carried


def test_envmap_alias_tables_are_a_valid_alias_method():
    rng = np.random.RandomState(0)
    w, h = 32, 16
    img = rng.rand(h, w, 3).astype(np.float32) ** 4
    img[3, 5] = 500.0
    e = host.envmap_from_rgb(w, h, img)
    n = w * h
    assert abs(e.pdf.sum() / n - 1.0) < 1e-3                       # pdf is pre-multiplied by n (step-function pdf)
    assert (e.prob >= 0).all() and (e.prob <= 1.0 + 1e-6).all() and (e.alias >= 0).all() and (e.alias < n).all()
    # the alias method must reproduce pdf/n: P(i) = (prob[i] + sum_{j: alias[j]==i} (1 - prob[j])) / n
    mass = e.prob.astype(np.float64).copy()
    np.add.at(mass, e.alias, 1.0 - e.prob.astype(np.float64))
    assert np.allclose(mass, e.pdf, rtol=2e-3, atol=2e-3)
    assert e.pdf.reshape(h, w)[3, 5] == e.pdf.max()


def _env_tables_by_the_reference_rules(img):
    """EnvironmentMap::computeProbabilities restated in numpy / plain Python (src/envmap.cpp:31-114): luminance x sin(theta) per texel in
    fp32 (libm sinf, as std::sin(float) resolves to), the integral as a sequential fp32 sum of scalar / n, pdf = scalar / I (1 / n when I is
    0), then Vose's alias method with two LIFO stacks split at p < 1, the large entry re-filed with (g + l) - 1."""
    import ctypes
    sinf = ctypes.CDLL("libm.so.6").sinf
    sinf.restype, sinf.argtypes = ctypes.c_float, [ctypes.c_float]
    f32 = np.float32
    h, w, _ = img.shape
    n = w * h
    pi = f32(3.14159265358979323846)
    sin_th = np.array([sinf(f32(pi * f32(v + 0.5)) / f32(h)) for v in range(h)], f32)
    r, g, b = img[..., 0].astype(f32), img[..., 1].astype(f32), img[..., 2].astype(f32)
    lum = (f32(0.212671) * r + f32(0.715160) * g) + f32(0.072169) * b
    scalars = (lum * sin_th[:, None]).astype(f32).reshape(-1)
    integral = np.cumsum(scalars / f32(n), dtype=f32)[-1]                    # I += scalars[i] / (w * h), in index order
    pdf = np.full(n, f32(1.0) / f32(n), f32) if integral == 0 else (scalars / integral).astype(f32)
    prob, alias = np.ones(n, f32), np.arange(n, dtype=np.int32)
    small = [(p_, i) for i, p_ in enumerate(pdf) if p_ < 1.0]
    large = [(p_, i) for i, p_ in enumerate(pdf) if not (p_ < 1.0)]
    while small and large:
        (pl, il), (pg, ig) = small.pop(), large.pop()
        prob[il], alias[il] = pl, ig
        pg2 = f32(f32(pg + pl) - f32(1.0))
        (small if pg2 < 1.0 else large).append((pg2, ig))
    return pdf, prob, alias


@pytest.mark.parametrize("w,h,seed", [(48, 24, 0), (64, 32, 1), (7, 5, 2)])
def test_envmap_tables_match_an_independent_restatement(w, h, seed):
    """host/envmap.cpp's pdf / probability / alias tables, bit for bit, against a second restatement of the reference's algorithm written
    in numpy (the reference's envmap.cpp cannot be built here: utils.h -> glad).  Entries whose probability ends at 1 never consult
    their alias; the reference leaves it uninitialised there, so it is compared only where prob < 1."""
    rng = np.random.RandomState(seed)
    img = (rng.rand(h, w, 3).astype(np.float32) ** 4) * np.float32(3.0)
    img[h // 3, w // 5] = 500.0
    img[h - 1, :] = 0.0                                                        # a black row
    e = host.envmap_from_rgb(w, h, img)
    pdf, prob, alias = _env_tables_by_the_reference_rules(img)
    assert np.array_equal(e.pdf.view(np.uint32), pdf.view(np.uint32))
    assert np.array_equal(e.prob.view(np.uint32), prob.view(np.uint32))
    used = prob < 1.0
    assert np.array_equal(e.alias[used], alias[used])


def test_envmap_all_black_falls_back_to_uniform():
    e = host.envmap_from_rgb(8, 4, np.zeros((4, 8, 3), np.float32))
    assert np.allclose(e.pdf, 1.0 / 32.0) and (e.prob <= 1.0).all()   # reference: src/envmap.cpp:62-63


@needs_ref_assets
@pytest.mark.ref
def test_hdr_reader_matches_reference_rgbe():
    """Our Radiance .hdr reader vs the reference's own src/rgbe/rgbe.cpp (compiled in oracle/_ref)."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    L = ob.ref_lib()
    path = (REF + "/env_maps/night.hdr").encode()
    w, h = C.c_int(), C.c_int()
    assert L.ref_read_hdr(path, C.byref(w), C.byref(h), None) == 0
    assert (w.value, h.value) == (512, 256)
    ref = np.zeros(w.value * h.value * 3, np.float32)
    assert L.ref_read_hdr(path, C.byref(w), C.byref(h), ref.ctypes.data_as(C.c_void_p)) == 0
    e = host.load_envmap(REF + "/env_maps/night.hdr")
    assert (e.w, e.h) == (512, 256) and np.array_equal(e.rgb, ref)
    assert abs(float(e.pdf[0]) - 0.00115893) < 1e-7                # value recorded in SURVEY 8(c) from the reference's EnvironmentMap


def test_procedural_scenes_are_deterministic_and_sized():
    a = host.generate_scene("kitchen", 40000, 42)
    b = host.generate_scene("kitchen", 40000, 42)
    assert a.tris.tobytes() == b.tris.tobytes() and a.materials.tobytes() == b.materials.tobytes()
    assert 0.7 * 40000 < a.tris.size < 1.3 * 40000
    types = a.materials["type"][1:]
    assert (types == wire.BXDF.DIFFUSE).sum() == 57 and (types == wire.BXDF.GLOSSY).sum() == 21    # Country-Kitchen.mtl mix
    assert (types == wire.BXDF.IDEAL_REFLECTION).sum() == 8 and (types == wire.BXDF.GGX_ROUGH_REFLECTION).sum() == 6
    assert (types == wire.BXDF.IDEAL_DIELECTRIC).sum() == 4 and a.texdesc.size == 17
    c = host.generate_scene("conference", 30000, 43)
    t = c.materials["type"][1:]
    assert abs((t == wire.BXDF.GGX_ROUGH_REFLECTION).mean() - 0.5) < 1e-6


def test_math_contract_accuracy():
    """include/flx_math.h vs numpy (float64): the gap the oracle-vs-reference tolerance has to cover."""
    L = ob.lib()
    f = np.vectorize(lambda fn, a, b=0.0: L.orc_math(fn, float(a), float(b)))
    x = np.linspace(-3.2, 6.4, 4001).astype(np.float32)
    assert np.abs(f(0, x) - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(f(1, x) - np.cos(x.astype(np.float64))).max() < 3e-7
    y = np.linspace(-1, 1, 2001).astype(np.float32)
    assert np.abs(f(4, y) - np.arccos(y.astype(np.float64))).max() < 6e-7
    a, b = np.random.RandomState(1).uniform(-2, 2, (2, 3000)).astype(np.float32)
    assert np.abs(f(3, a, b) - np.arctan2(a.astype(np.float64), b.astype(np.float64))).max() < 6e-7
    c = np.linspace(1e-3, 4, 3000).astype(np.float32)
    for e in (2.2, 1.0 / 2.2):
        ref = c.astype(np.float64) ** np.float64(np.float32(e))
        assert (np.abs(f(5, c, e) - ref) / ref).max() < 2e-6
    assert abs(L.orc_math(2, float(np.float32(np.pi / 6)), 0.0) - np.tan(np.float32(np.pi / 6))) < 2e-7
    # hash RNG known answers (reference: src/random.cl:7-15, computed by hand from the definition)
    def h(s):
        s = ((s ^ 61) ^ (s >> 16)) & 0xffffffff; s = (s * 9) & 0xffffffff; s ^= s >> 4
        s = (s * 0x27d4eb2d) & 0xffffffff; s ^= s >> 15
        return s
    for s in (0, 1, 61, 2**32 - 1, 123456789):
        assert L.orc_hash(s) == h(s)


@pytest.mark.parametrize("mode,bits", [("RGB", 8), ("RGBA", 8), ("L", 8), ("LA", 8), ("P", 8), ("I;16", 16)])
def test_png_decoder_matches_pillow(tmp_path, mode, bits):
    """PNG decode (zlib inflate + scanline filters) against Pillow for every supported colour type; origin lower-left."""
    from PIL import Image
    rng = np.random.RandomState(5)
    w, h = 37, 23                       # odd sizes exercise every filter type's edge handling
    if mode == "I;16":
        arr = rng.randint(0, 65536, (h, w)).astype(np.uint16)
        im = Image.fromarray(arr, "I;16")
        expect = np.stack([(arr >> 8).astype(np.uint8)] * 3 + [np.full((h, w), 255, np.uint8)], -1)
    else:
        base = (np.linspace(0, 255, w)[None, :, None] * np.ones((h, 1, 4)) + rng.randint(0, 40, (h, w, 4))).clip(0, 255).astype(np.uint8)
        im = Image.fromarray(base, "RGBA").convert(mode)
        expect = np.asarray(im.convert("RGBA"))
    path = str(tmp_path / f"t_{mode.replace(';', '')}.png")
    im.save(path)
    got = host.load_png(path)
    assert got.shape == (h, w, 4) and np.array_equal(got, expect[::-1])


def test_png_decoder_rejects_garbage(tmp_path):
    p = tmp_path / "bad.png"
    p.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\x00" * 64)
    with pytest.raises(RuntimeError):
        host.load_png(str(p))


@needs_ref_assets
def test_obj_loader_resolves_png_textures():
    d = host.load_scene(REF + "/egyptcat/egyptcat.obj")
    assert d.texdesc.size == 1 and tuple(d.texdesc[0]) == (0, 1024, 1024) and d.texdata.size == 1024 * 1024 * 4
    assert d.materials[1]["map_Kd"] == 0 and d.materials[2]["map_Kd"] == -1        # `map_Kd EgyptCat.png` on material egyptcat
    from PIL import Image
    ref = np.asarray(Image.open(REF + "/egyptcat/EgyptCat.png").convert("RGBA"))[::-1]
    assert np.array_equal(d.texdata.reshape(1024, 1024, 4), ref)


def _jpeg_image(w, h, mode, seed):
    """Smooth gradients + texture + hard edges + noise: exercises DC prediction, long zero runs, EOB runs and ringing."""
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 127 + 120 * np.sin(x / 9.0) * np.cos(y / 13.0)
    g = 255.0 * x / max(1, w - 1)
    b = 255.0 * ((x // 8 + y // 8) % 2)
    img = np.stack([r, g, b], -1) + rng.normal(0, 6, (h, w, 3))
    img[h // 3: h // 3 + 5, :, :] = 255
    arr = img.clip(0, 255).astype(np.uint8)
    return arr[..., 1] if mode == "L" else arr


@pytest.mark.parametrize("size", [(64, 48), (37, 23), (17, 9), (8, 8), (1, 1), (3, 130)])
@pytest.mark.parametrize("kw", [dict(quality=90, subsampling=0), dict(quality=75, subsampling=1), dict(quality=60, subsampling=2),
                                dict(quality=95, subsampling=2, progressive=True), dict(quality=50, subsampling=0, progressive=True, optimize=True),
                                dict(quality=85, subsampling=2, optimize=True, restart_marker_blocks=3), dict(quality=30, subsampling=1, restart_marker_rows=1),
                                dict(quality=100, subsampling=0), dict(quality=5, subsampling=2)],
                         ids=["444", "422", "420", "420-progressive", "444-progressive-opt", "420-restart-blocks", "422-restart-rows", "q100", "q5"])
def test_jpeg_decoder_is_bit_identical_to_libjpeg(size, kw):
    """SURVEY 8(f) N2.  host/jpeg.cpp restates libjpeg's default decode path (ISLOW IDCT, fancy upsampling, fixed-point
    YCbCr->RGB); Pillow decodes with libjpeg-turbo, whose output is defined to equal libjpeg's: every byte must match."""
    from PIL import Image
    import io
    w, h = size
    for mode in ("RGB", "L"):
        arr = _jpeg_image(w, h, mode, seed=w * 131 + h)
        buf = io.BytesIO()
        Image.fromarray(arr, mode).save(buf, "JPEG", **kw)
        data = buf.getvalue()
        expect = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        got = host.decode_jpeg(data)
        assert got.shape == expect.shape
        assert np.array_equal(got, expect), f"{mode} {kw}: {(got != expect).mean():.4%} of the bytes differ"


def test_jpeg_loader_origin_and_errors(tmp_path):
    from PIL import Image
    arr = _jpeg_image(40, 24, "RGB", 3)
    path = str(tmp_path / "t.jpg")
    Image.fromarray(arr, "RGB").save(path, "JPEG", quality=92)
    ref = np.asarray(Image.open(path).convert("RGBA"))
    got = host.load_texture(path)                                   # dispatch on the file signature, lower-left origin, alpha 255
    assert got.shape == (24, 40, 4) and np.array_equal(got, ref[::-1])
    for bad in (b"\xff\xd8\xff\xd9", b"\xff\xd8" + b"\x00" * 40, b"\xff\xd8\xff\xc9\x00\x0b\x08\x00\x08\x00\x08\x01\x01\x11\x00"):   # no frame; garbage; arithmetic coding
        with pytest.raises(RuntimeError):
            host.decode_jpeg(bad)
    data = open(path, "rb").read()
    trunc = host.decode_jpeg(data[: len(data) * 2 // 3])           # truncated entropy data: zero-filled like libjpeg, no crash
    assert trunc.shape == (24, 40, 3)


@needs_ref_assets
def test_jpeg_decoder_on_the_country_kitchen_textures():
    """The reference's own JPEG assets (baseline 4:4:4, baseline 4:2:0, progressive; Exif / JFIF / Adobe headers)."""
    from PIL import Image
    import glob
    files = sorted(glob.glob(REF + "/country_kitchen/textures/*.jpg"))
    assert len(files) >= 10
    for f in files:
        expect = np.asarray(Image.open(f).convert("RGB"))
        assert np.array_equal(host.decode_jpeg(open(f, "rb").read()), expect), f


def test_obj_loader_resolves_jpeg_textures(tmp_path):
    from PIL import Image
    arr = _jpeg_image(32, 16, "RGB", 9)
    Image.fromarray(arr, "RGB").save(str(tmp_path / "wood.jpg"), "JPEG", quality=80, subsampling=2)
    (tmp_path / "m.mtl").write_text("newmtl wood\nKd 0.5 0.5 0.5\nmap_Kd wood.jpg\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl wood\nf 1/1 2/2 3/3\n")
    d = host.load_scene(str(tmp_path / "m.obj"))
    assert d.texdesc.size == 1 and tuple(d.texdesc[0])[1:] == (32, 16) and d.materials[1]["map_Kd"] == 0
    ref = np.asarray(Image.open(str(tmp_path / "wood.jpg")).convert("RGBA"))[::-1]
    assert np.array_equal(d.texdata.reshape(16, 32, 4), ref)


# ---------------------------------------------------------------- PBRT ingest (parity unpinned, see host/pbrt.cpp)
PBRT_SCENE = '''
# a hand-built pbrt-v3 scene exercising the subset the reference's loader maps (src/scene.cpp:574-813)
LookAt 0 -6 2   0 0 1   0 0 1          # Z-up camera
Camera "perspective" "float fov" [ 45 ]
Film "image" "integer xresolution" [ 64 ] "integer yresolution" [ 48 ] "string filename" "out.png"
Sampler "halton" "integer pixelsamples" 4
WorldBegin
LightSource "infinite" "rgb L" [ 1 1 1 ]
Texture "wood" "spectrum" "imagemap" "string filename" "wood.jpg"
Texture "scaled" "float" "scale" "float tex1" 2 "float tex2" 3
MakeNamedMaterial "gold" "string type" "metal" "rgb eta" [ 0.2 0.9 1.1 ] "rgb k" [ 3 2.3 1.8 ] "float roughness" 0.2
MakeNamedMaterial "pane" "string type" "glass" "float index" 1.33 "rgb Kt" [ .9 .9 .9 ]
AttributeBegin
  Material "matte" "texture Kd" "wood"
  Translate 1 2 3
  Shape "trianglemesh" "integer indices" [ 0 1 2  0 2 3 ] "point P" [ 0 0 0  1 0 0  1 1 0  0 1 0 ] "float uv" [ 0 0  1 0  1 1  0 1 ]
AttributeEnd
AttributeBegin
  NamedMaterial "gold"
  Scale 2 2 2
  Rotate 90 0 0 1
  Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ 1 0 0  0 1 0  0 0 1 ] "normal N" [ 0 0 1  0 0 1  0 0 1 ]
  Shape "sphere" "float radius" 1        # skipped by the reference too
AttributeEnd
ObjectBegin "leaf"
  Material "plastic" "rgb Kd" [ .1 .6 .1 ] "rgb Ks" [ .3 .3 .3 ] "float roughness" 0.05
  Shape "plymesh" "string filename" "leaf.ply"
ObjectEnd
AttributeBegin
  Translate 10 0 0
  ObjectInstance "leaf"
  Translate 0 5 0
  ObjectInstance "leaf"
AttributeEnd
NamedMaterial "pane"
Shape "trianglemesh" "point P" [ 0 0 5  1 0 5  0 1 5 ]
Material "mirror"
Include "more.pbrt"
WorldEnd
'''


def _write_ply(path, binary):
    verts = np.array([[0, 0, 0, 0, 0, 1, 0, 0], [1, 0, 0, 0, 0, 1, 1, 0], [1, 1, 0, 0, 0, 1, 1, 1], [0, 1, 0, 0, 0, 1, 0, 1]], np.float32)
    hdr = ("ply\nformat %s 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\n"
           "property float nz\nproperty float u\nproperty float v\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n")
    with open(path, "wb") as f:
        if binary:
            f.write((hdr % "binary_little_endian").encode())
            f.write(verts.tobytes()); f.write(bytes([4])); f.write(np.array([0, 1, 2, 3], np.int32).tobytes())
        else:
            f.write((hdr % "ascii").encode())
            for v in verts:
                f.write((" ".join("%g" % x for x in v) + "\n").encode())
            f.write(b"4 0 1 2 3\n")


@pytest.mark.parametrize("binary_ply", [False, True])
def test_pbrt_scene_ingest(tmp_path, binary_ply):
    from PIL import Image
    Image.fromarray(_jpeg_image(16, 8, "RGB", 2), "RGB").save(str(tmp_path / "wood.jpg"), "JPEG")
    _write_ply(str(tmp_path / "leaf.ply"), binary_ply)
    (tmp_path / "more.pbrt").write_text('Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 9  1 0 9  0 1 9]\n')
    (tmp_path / "scene.pbrt").write_text(PBRT_SCENE)
    d = host.load_scene(str(tmp_path / "scene.pbrt"))
    t, m = d.tris, d.materials
    # world shapes in file order, then the instances: quad (2), rotated tri (1), glass tri (1), included mirror tri (1), 2 x leaf quad (2 x 2)
    assert t.size == 9 and list(t["matId"]) == [1, 1, 2, 3, 4, 5, 5, 5, 5]
    assert m.size == 6                                                       # default + 5 in order of first use
    assert [int(x) for x in m["type"][1:]] == [wire.BXDF.DIFFUSE, wire.BXDF.GGX_ROUGH_REFLECTION, wire.BXDF.IDEAL_DIELECTRIC,
                                              wire.BXDF.IDEAL_REFLECTION, wire.BXDF.GLOSSY]
    P = lambda i, v: np.array([t[i][v]["p"][k] for k in "xyz"])
    assert np.allclose(P(0, "v0"), [1, 2, 3]) and np.allclose(P(0, "v2"), [2, 3, 3])                    # Translate
    assert np.allclose(P(2, "v0"), [0, 2, 0], atol=1e-5) and np.allclose(P(2, "v1"), [-2, 0, 0], atol=1e-5)   # Scale then Rotate 90 about z
    n2 = np.array([t[2]["v0"]["n"][k] for k in "xyz"])
    assert np.allclose(n2, [0, 0, 0.5], atol=1e-6)                                                      # inverse transpose, not renormalised
    n3 = np.array([t[3]["v0"]["n"][k] for k in "xyz"])
    assert np.allclose(n3, [0, 0, 1])                                                                   # no normals: flat normal
    assert np.allclose(P(5, "v0"), [10, 0, 0]) and np.allclose(P(7, "v0"), [10, 5, 0])                 # instancing, CTM accumulates
    assert np.allclose([t[6]["v1"]["t"]["x"], t[6]["v1"]["t"]["y"]], [1, 1])                           # quad fan 0-2-3 keeps the uvs
    # materials: the reference's mapping
    assert m[1]["map_Kd"] == 0 and d.texdesc.size == 1 and tuple(d.texdesc[0])[1:] == (16, 8)
    assert np.isclose(m[2]["Ni"], (0.2 + 0.9 + 1.1) / 3) and np.isclose(m[2]["Ns"], (1 - 0.2) * 5000) and np.allclose([m[2]["Ks"][k] for k in "xyz"], [3, 2.3, 1.8])
    assert np.isclose(m[3]["Ni"], 1.33) and np.allclose([m[3]["Ks"][k] for k in "xyz"], 0.9)
    assert np.allclose([m[4]["Ks"][k] for k in "xyz"], 0.9)
    assert np.isclose(m[5]["Ns"], (1 - 0.05) * 5000) and np.isclose(m[5]["Ni"], 1.5) and np.allclose([m[5]["Kd"][k] for k in "xyz"], [.1, .6, .1])
    assert d.world_up == (0.0, 0.0, 1.0)                                                                # Z-up camera frame
    host.build_bvh(d, "sbvh")                                                                           # and it feeds the rest of the host path
    assert d.nodes.size >= 1


def test_pbrt_errors(tmp_path):
    (tmp_path / "a.pbrt").write_text('WorldBegin\nShape "sphere"\nWorldEnd\n')
    with pytest.raises(RuntimeError, match="without triangle"):
        host.load_scene(str(tmp_path / "a.pbrt"))
    (tmp_path / "b.pbrt").write_text('WorldBegin\nAttributeEnd\n')
    with pytest.raises(RuntimeError, match="unmatched"):
        host.load_scene(str(tmp_path / "b.pbrt"))
    (tmp_path / "c.pbrt").write_text('WorldBegin\nShape "plymesh" "string filename" "missing.ply"\n')
    with pytest.raises(RuntimeError, match="cannot open"):
        host.load_scene(str(tmp_path / "c.pbrt"))


@pytest.mark.parametrize("kind", ["kitchen", "courtyard"])
def test_pbrt_round_trip_of_a_procedural_scene(tmp_path, kind):
    """Procedural scene -> pbrt-v3 files (fluctus_amd/pbrt_export.py) -> Scene::loadPBRTModel: same triangles bit for bit,
    materials back through the reference's PBRT mapping (binary PLY meshes, every material class it maps)."""
    from fluctus_amd import pbrt_export
    d = host.generate_scene(kind, 12000, 5)
    path, skipped = pbrt_export.export(d, str(tmp_path), "s")
    r = host.load_scene(path)
    assert r.tris.size == d.tris.size
    # the PBRT path numbers materials by first use and groups triangles by mesh: compare per material group
    order = []
    for mid in d.tris["matId"]:
        if int(mid) not in order:
            order.append(int(mid))
    off = 0
    for new_id, mid in enumerate(order):
        src = d.tris[d.tris["matId"] == mid]
        dst = r.tris[off:off + src.size]
        off += src.size
        for v in ("v0", "v1", "v2"):
            assert np.array_equal(src[v]["p"], dst[v]["p"]) and np.array_equal(src[v]["n"], dst[v]["n"])
            assert np.array_equal(src[v]["t"]["x"], dst[v]["t"]["x"]) and np.array_equal(src[v]["t"]["y"], dst[v]["t"]["y"])
        want = 0 if mid == 0 else new_id + (0 if 0 in order[:new_id + 1] else 1)
        assert (dst["matId"] == dst["matId"][0]).all()
        if mid == 0 or mid in skipped:
            continue
        a, b = d.materials[mid], r.materials[int(dst["matId"][0])]
        assert int(a["type"]) == int(b["type"])
        for f in ("Kd", "Ks"):
            if int(a["type"]) in (wire.BXDF.DIFFUSE,) and f == "Ks":
                continue
            if int(a["type"]) not in (wire.BXDF.DIFFUSE, wire.BXDF.GLOSSY) and f == "Kd":
                continue
            assert np.allclose([a[f][k] for k in "xyz"], [b[f][k] for k in "xyz"], rtol=1e-6), (mid, f)
        if int(a["type"]) in (wire.BXDF.GLOSSY, wire.BXDF.GGX_ROUGH_REFLECTION):
            assert np.isclose(a["Ns"], b["Ns"], rtol=1e-4, atol=1e-2)
        if int(a["type"]) in (wire.BXDF.GLOSSY, wire.BXDF.GGX_ROUGH_REFLECTION, wire.BXDF.IDEAL_DIELECTRIC):
            assert np.isclose(a["Ni"], b["Ni"], rtol=1e-6)
    assert off == r.tris.size


# ---------------------------------------------------------------- on-disk caches in the reference's formats
def test_xxh64_matches_the_xxhash_library():
    import xxhash
    rng = np.random.RandomState(1)
    for n in (0, 1, 3, 4, 7, 8, 15, 31, 32, 33, 63, 64, 100, 4097):
        data = rng.randint(0, 256, n).astype(np.uint8).tobytes()
        for seed in (0, 1, 2**63 + 5):
            assert host.xxh64(data, seed) == xxhash.xxh64(data, seed=seed).intdigest(), (n, seed)


@pytest.mark.parametrize("mode", ["sbvh", "sah"])
def test_hierarchy_cache_file_is_the_reference_format(tmp_path, mode):
    """src/bvh.cpp:102-192: u32 #indices, indices, u32 count, 33 bytes per node {6 floats, u32 iStart|rightChild, i32 parent, u8 nPrims}."""
    d = host.generate_scene("conference", 3000, 1)
    host.build_bvh(d, mode)
    path = str(tmp_path / "hierarchy_1.bin")
    host.bvh_export(d, path, mode)
    raw = open(path, "rb").read()
    ni = int(np.frombuffer(raw, "<u4", 1)[0])
    assert ni == d.indices.size and np.array_equal(np.frombuffer(raw, "<u4", ni, 4), d.indices)
    nn = int(np.frombuffer(raw, "<u4", 1, 4 + 4 * ni)[0])
    assert nn == d.nodes.size and len(raw) == 8 + 4 * ni + 33 * nn
    rec = np.frombuffer(raw, np.dtype([("box", "<f4", 6), ("istart", "<u4"), ("parent", "<i4"), ("nprims", "u1")]), nn, 8 + 4 * ni)
    assert np.array_equal(rec["istart"], d.nodes["iStartOrRight"]) and np.array_equal(rec["parent"], d.nodes["parent"]) and np.array_equal(rec["nprims"], d.nodes["nPrims"])
    assert np.array_equal(rec["box"][:, 0], d.nodes["bmin"]["x"]) and np.array_equal(rec["box"][:, 5], d.nodes["bmax"]["z"])
    nodes, idx = host.bvh_import(path)
    assert np.array_equal(idx, d.indices)
    for f in ("parent", "iStartOrRight", "nPrims"):
        assert np.array_equal(nodes[f], d.nodes[f])
    assert np.array_equal(nodes["bmin"], d.nodes["bmin"]) and np.array_equal(nodes["bmax"], d.nodes["bmax"])
    # a file as the REFERENCE writes it: the count field holds #indices (src/bvh.cpp:185); every node must still come back
    bad = bytearray(raw); bad[4 + 4 * ni: 8 + 4 * ni] = np.uint32(ni).tobytes()
    p2 = str(tmp_path / "ref.bin"); open(p2, "wb").write(bytes(bad))
    nodes2, _ = host.bvh_import(p2)
    assert nodes2.size == nn
    open(p2, "wb").write(raw[:-5])                                    # truncated file: refused
    with pytest.raises(RuntimeError):
        host.bvh_import(p2)


def test_bench_workload_hierarchy_cache_is_transparent(tmp_path, monkeypatch):
    """bench.py caches the hierarchy of its procedural scenes in the reference's file format: a cached run must hand the device
    exactly the arrays and parameters a fresh build does."""
    import bench
    monkeypatch.setenv("FLX_BVH_CACHE", str(tmp_path))
    monkeypatch.setenv("FLX_BENCH_TRIS", "20000")
    d1, p1, _ = bench.build_workload(name="kitchen")
    assert len(list(tmp_path.iterdir())) == 1
    d2, p2, _ = bench.build_workload(name="kitchen")
    assert np.array_equal(d1.nodes, d2.nodes) and np.array_equal(d1.indices, d2.indices)
    assert d1.world_radius == d2.world_radius and np.array_equal(p1, p2)


def _tinyobj_expected(path):
    """What Scene::loadObjWithMaterials makes of an OBJ: the reference's VENDORED parser (include/tiny_obj_loader.h, run through
    oracle/ref/tinyobj_driver.cpp) + the conventions of src/scene.cpp:13-26, 171-189, 236-301 restated here in numpy:
    material 0 = built-in default, matId = tinyobj's id + 1, a triangle with any corner lacking a normal gets the flat normal
    normalize(cross(p1 - p0, p2 - p0)) on all three corners, missing texcoords are (0, 0), `shader` MTL key -> BSDF type."""
    import ctypes as C
    from oracle.binding import ref_lib
    L = ref_lib()
    h = C.c_void_p()
    folder = path[:path.rfind("/") + 1]
    assert L.ref_tinyobj_load(path.encode(), folder.encode(), C.byref(h)) == 0
    try:
        nt, nm, hn, ht = C.c_uint64(), C.c_uint64(), C.c_int(), C.c_int()
        L.ref_tinyobj_counts(h, C.byref(nt), C.byref(nm), C.byref(hn), C.byref(ht))
        n, m = nt.value, nm.value
        pos = np.zeros((n, 3, 3), np.float32); nrm = np.zeros((n, 3, 3), np.float32); uv = np.zeros((n, 3, 2), np.float32)
        nidx = np.zeros((n, 3), np.int32); tidx = np.zeros((n, 3), np.int32); mid = np.zeros(n, np.int32)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        assert L.ref_tinyobj_faces(h, P(pos), P(nrm), P(nidx), P(uv), P(tidx), P(mid)) == 0
        vals = np.zeros((max(m, 1), 11), np.float32); names = np.zeros((max(m, 1), 4, 256), np.uint8)
        L.ref_tinyobj_materials(h, P(vals), P(names))
    finally:
        L.ref_tinyobj_free(h)
    # src/scene.cpp:252-276
    no_normal = (nidx < 0) | (not hn.value)
    flat = no_normal.any(1)
    e1, e2 = pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]
    cr = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2], e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], 1).astype(np.float32)
    ln = np.sqrt((cr * cr).sum(1, dtype=np.float32)).astype(np.float32)
    with np.errstate(all="ignore"):
        fn = (cr / ln[:, None]).astype(np.float32)
    nrm = np.where(flat[:, None, None], fn[:, None, :], np.where(no_normal[:, :, None], 0.0, nrm)).astype(np.float32)
    uv = np.where(((tidx < 0) | (not ht.value))[:, :, None], 0.0, uv).astype(np.float32)
    shader = {"diffuse": wire.BXDF.DIFFUSE, "glossy": wire.BXDF.GLOSSY, "rough_reflection": wire.BXDF.GGX_ROUGH_REFLECTION,
              "ideal_reflection": wire.BXDF.IDEAL_REFLECTION, "rough_dielectric": wire.BXDF.GGX_ROUGH_DIELECTRIC,
              "ideal_dielectric": wire.BXDF.IDEAL_DIELECTRIC, "emissive": wire.BXDF.EMISSIVE}
    mats = []
    for i in range(m):
        s = [bytes(names[i, k]).split(b"\0")[0].decode() for k in range(4)]
        mats.append(dict(Kd=vals[i, 0:3], Ks=vals[i, 3:6], Ke=vals[i, 6:9], Ns=vals[i, 9], Ni=vals[i, 10], tex=s[:3], type=shader.get(s[3], wire.BXDF.DIFFUSE)))
    return pos, nrm, uv, mid + 1, flat, mats


@needs_ref_assets
@pytest.mark.ref
@pytest.mark.parametrize("rel", ["egyptcat/egyptcat.obj", "gold_rings/gold_rings_bark.obj", "psor/psor-cube.obj"])
def test_obj_loader_matches_the_reference_parser_and_conventions(rel):
    """SURVEY A13, pinned against third-party code the reference really runs: host/scene.cpp's own OBJ + MTL parser vs tinyobj
    (compiled from the reference's vendored header) on every OBJ the reference checkout ships, with scene.cpp's conventions on top:
    triangle order and fan triangulation, positions, per-corner normals / the flat-normal rule, texcoords, matId + 1, and per
    material Kd / Ks / Ke / Ns / Ni / shader type / which texture slots are set -- all exact."""
    from oracle.binding import ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built")
    path = REF + "/" + rel
    if not os.path.exists(path):
        pytest.skip(rel + " not in the checkout")
    pos, nrm, uv, mid, flat, mats = _tinyobj_expected(path)
    d = host.load_scene(path)
    assert d.tris.size == pos.shape[0]
    for vi, v in enumerate(("v0", "v1", "v2")):
        got_p = np.stack([d.tris[v]["p"][k] for k in "xyz"], 1)
        got_n = np.stack([d.tris[v]["n"][k] for k in "xyz"], 1)
        got_t = np.stack([d.tris[v]["t"][k] for k in "xy"], 1)
        assert np.array_equal(got_p, pos[:, vi]), (rel, v, "positions")
        assert np.array_equal(got_t, uv[:, vi]), (rel, v, "texcoords")
        # stored normals exact; flat normals: same formula, fp32 -- allow the last ulp of the normalisation (division vs reciprocal)
        assert np.array_equal(got_n[~flat], nrm[~flat, vi]), (rel, v, "vertex normals")
        assert np.allclose(got_n[flat], nrm[flat, vi], rtol=0, atol=2e-7, equal_nan=True), (rel, v, "flat normals")
    assert np.array_equal(d.tris["matId"], mid), (rel, "matId = tinyobj id + 1")
    assert d.materials.size == len(mats) + 1                                     # + the built-in default at index 0
    for i, m in enumerate(mats):
        g = d.materials[i + 1]
        for key in ("Kd", "Ks", "Ke"):
            assert np.array_equal(np.array([g[key]["x"], g[key]["y"], g[key]["z"]], np.float32), m[key]), (rel, i, key)
        assert g["Ns"] == m["Ns"] and g["Ni"] == m["Ni"] and int(g["type"]) == int(m["type"]), (rel, i)
        for slot, name in zip(("map_Kd", "map_Ks", "map_N"), m["tex"]):
            # the reference imports a texture only if the file exists (tryImportTexture, src/scene.cpp:304-321)
            exists = bool(name) and os.path.exists(path[:path.rfind("/") + 1] + name.replace("\\", "/"))
            assert (int(g[slot]) >= 0) == exists, (rel, i, slot, name)


@needs_ref_assets
@pytest.mark.ref
@pytest.mark.parametrize("folder,mtl", [("country_kitchen", "Country-Kitchen.mtl"), ("conference", "conference.mtl"), ("luxball", "luxball.mtl")])
def test_mtl_parser_matches_the_reference_parser_on_the_material_libraries(tmp_path, folder, mtl):
    """The three material libraries whose OBJ is a missing blob in the checkout (Country Kitchen: 96 materials, 17 map_Kd + a map_Bump;
    conference: 34; luxball): a one-triangle OBJ that pulls the library in, through tinyobj and through host/scene.cpp."""
    from oracle.binding import ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built")
    src = REF + "/" + folder
    if not os.path.exists(src + "/" + mtl):
        pytest.skip(mtl + " not in the checkout")
    import shutil
    shutil.copy(src + "/" + mtl, tmp_path / mtl)
    if os.path.isdir(src + "/textures"):
        os.symlink(src + "/textures", tmp_path / "textures")
    names = [l.split(None, 1)[1].strip() for l in open(src + "/" + mtl, errors="replace") if l.startswith("newmtl ")]
    with open(tmp_path / "probe.obj", "w") as f:
        f.write(f"mtllib {mtl}\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\n")
        for k, nm in enumerate(names):
            f.write(f"usemtl {nm}\nf 1/1 2/2 3/3\n")
    pos, nrm, uv, mid, flat, mats = _tinyobj_expected(str(tmp_path / "probe.obj"))
    d = host.load_scene(str(tmp_path / "probe.obj"))
    assert len(mats) == len(names) and d.materials.size == len(mats) + 1 and d.tris.size == len(names)
    assert np.array_equal(d.tris["matId"], mid)
    ntex = 0
    for i, m in enumerate(mats):
        g = d.materials[i + 1]
        for key in ("Kd", "Ks", "Ke"):
            assert np.array_equal(np.array([g[key]["x"], g[key]["y"], g[key]["z"]], np.float32), m[key]), (mtl, names[i], key)
        assert g["Ns"] == m["Ns"] and g["Ni"] == m["Ni"] and int(g["type"]) == int(m["type"]), (mtl, names[i])
        for slot, name in zip(("map_Kd", "map_Ks", "map_N"), m["tex"]):
            exists = bool(name) and os.path.exists(str(tmp_path) + "/" + name.replace("\\", "/"))
            assert (int(g[slot]) >= 0) == exists, (mtl, names[i], slot, name)
            ntex += exists
    if folder == "country_kitchen":
        assert len(names) == 96 and ntex >= 17


@pytest.mark.ref
@pytest.mark.parametrize("scene", ["teapot", "small", "conference-38k", "kitchen-30k"])
def test_sah_bvh_is_the_reference_builders_tree(tmp_path, scene):
    """SURVEY A12 / G8, for the builder that CAN be built here: the reference's own `class BVH` (src/bvh.cpp + src/bvhnode.cpp compiled
    unmodified into oracle/_ref/libfluctus_refbvh.so) builds the tree with SplitMode::SAH and writes it with its own BVH::exportTo;
    host/bvh.cpp's Mode::SAH must produce the SAME node array and index list, byte for byte, and host/bvh.cpp's importFrom must read the
    reference's file (the on-disk cache format of src/bvh.cpp:147-192, incl. its node-count quirk).  (src/sbvh.cpp, the builder
    Tracer::initHierarchy uses, includes progressview.hpp -> glad / GLFW / nanogui and cannot be built; our SBVH shares the sort,
    sweep and tie-break rules pinned here.)"""
    import ctypes as C
    from oracle.binding import refbvh_available, refbvh_lib
    if not refbvh_available():
        pytest.skip("oracle/_ref/libfluctus_refbvh.so not built (needs /root/reference)")
    if scene == "teapot":
        if not os.path.exists(REF + "/teapot.ply"):
            pytest.skip("teapot.ply not in the checkout")
        d = host.load_scene(REF + "/teapot.ply")
    elif scene == "small":
        d = common.small_mesh_scene(n=6)
    elif scene == "conference-38k":
        d = host.generate_scene("conference", 6000, 43)
    else:
        d = host.generate_scene("kitchen", 30000, 42)
    f = str(tmp_path / "ref_hierarchy.bin")
    assert refbvh_lib().ref_bvh_build_export(d.tris.ctypes.data_as(C.c_void_p), C.c_uint64(d.tris.size), 0, f.encode()) == 0
    rn, ri = host.bvh_import(f)
    host.build_bvh(d, "sah")
    assert rn.size == d.nodes.size and np.array_equal(ri, d.indices)
    assert np.array_equal(rn.view(np.uint8), d.nodes.view(np.uint8))


def test_night_env_fixture_tables_are_reproducible():
    """tests/golden/night_env.npz (scripts/make_envmap_fixture.py): the reference's night.hdr as pixels; bench.night_env() rebuilds the alias / pdf tables
    with host/envmap.cpp and checks them against the digest taken when the fixture was made; pdf[0] is the value SURVEY 8(c) records from the reference's own
    EnvironmentMap (0.00115893)."""
    import bench
    e = bench.night_env()
    assert (e.w, e.h) == (512, 256)
    assert abs(float(e.pdf[0]) - 0.00115893) < 1e-7
    assert e.prob.size == 512 * 256 and (e.prob <= 1.0).all() and (e.alias >= 0).all() and (e.alias < e.prob.size).all()
    if os.path.isdir(REF):            # build container: identical to loading the .hdr itself
        f = host.load_envmap(REF + "/env_maps/night.hdr")
        assert np.array_equal(f.rgb, e.rgb) and np.array_equal(f.prob, e.prob) and np.array_equal(f.alias, e.alias) and np.array_equal(f.pdf, e.pdf)
