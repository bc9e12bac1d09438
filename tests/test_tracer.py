"""The C++ headless Tracer (host loop of the reference's Tracer::update / runBenchmark over the C ABI)."""
import numpy as np
import pytest
import common
from fluctus_amd import host, wire, driver


def test_tracer_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from fluctus_amd.tracer import Tracer
    with pytest.raises(RuntimeError, match="no HIP device|cannot load"):
        Tracer(64, 64, 0, 1024)


@pytest.mark.gpu
def test_cpp_tracer_update_matches_oracle(tmp_path):
    from fluctus_amd.tracer import Tracer
    from oracle.binding import OracleContext
    w, h, n = 96, 64, 8192
    scene = "proc:conference:12000:43"
    t = Tracer(w, h, 0, n)
    t.set_option("extend_tree", 2)          # the reference's visit order: counters and image compared exactly below
    t.init(w, h, scene)
    p = t.params
    wire.look_at(p, (0.0, 1.2, 2.6), (0.0, 0.2, 0.0))
    p["maxBounces"], p["wfSeparateQueues"] = 6, 1
    t.params = p
    p = t.params
    d = host.generate_scene("conference", 12000, 43)
    host.build_bvh(d, "sbvh")
    assert abs(float(p["worldRadius"]) - d.world_radius) < 1e-6 and int(p["n_tris"]) == d.tris.size
    o = OracleContext(n, threads=8)
    o.upload_scene(d); o.set_params(p)
    cg = t.update()
    co = driver.first_frame(o, p, w * h)
    assert (cg == co).all()
    for _ in range(5):
        cg = t.update()
        co = driver.benchmark_iteration(o, w * h)
        assert (cg == co).all()
    pg, po = t.read_pixels(0), o.read_pixels(0)
    assert common.fb_close(pg, po)
    o.postprocess()
    assert np.allclose(t.read_pixels(1), o.read_pixels(1), rtol=4e-6, atol=1e-6)
    t.save_image(str(tmp_path / "a.ppm")); t.save_image(str(tmp_path / "a.pfm"))
    assert (tmp_path / "a.ppm").stat().st_size > w * h * 3 and (tmp_path / "a.pfm").stat().st_size > w * h * 12
    csv = t.run_benchmark(0.0, iterations=12)
    rows = csv.strip().split("\n")
    assert rows[0] == "scene;time;primary;extension;shadow;total;samples" and len(rows) >= 2
    vals = [float(x) for x in rows[-1].split(";")[1:]]
    assert vals[2] > 0 and abs(vals[4] - (vals[1] + vals[2] + vals[3])) <= 1e-4 * vals[4]   # total = primary+extension+shadow


@pytest.mark.gpu
def test_cpp_tracer_render_single_and_microkernel_update():
    """Tracer::renderSingle (src/tracer.cpp:95-187) and the MK branch of update()/runBenchmark (:268-299, :441-447)
    against the oracle driven by the same sequence: bit-identical images, exact spp, same ray statistics."""
    from fluctus_amd.tracer import Tracer
    from oracle.binding import OracleContext
    w, h = 80, 60
    n = w * h
    t = Tracer(w, h, 0, n)
    t.init(w, h, "proc:kitchen:9000:7")
    p = t.params
    wire.look_at(p, (0.0, 1.2, 2.6), (0.0, 0.2, 0.0))
    p["maxBounces"], p["useRoulette"] = 3, 1
    t.params = p
    d = host.generate_scene("kitchen", 9000, 7)
    host.build_bvh(d, "sbvh")
    o = OracleContext(n, threads=8)
    o.upload_scene(d)
    assert t.uses_wavefront
    t.render_single(5)
    assert not t.uses_wavefront and int(t.params["useRoulette"]) == 0
    p = t.params
    driver.render_single(o, p, 5)
    pg, po = t.read_pixels(0), o.read_pixels(0)
    assert (pg[:, 3] == 5).all() and np.array_equal(pg.view(np.uint32), po.view(np.uint32))
    so = o.mk_stats().astype(np.uint64)
    assert np.array_equal(t.stats(), so) and so[3] == 5 * n
    # interactive MK frames: params change -> iteration 0 = reset + 2-segment preview, then one state-machine step per frame
    p["maxBounces"] = 4
    t.params = p
    o.set_params(p)
    t.update()
    o.mk_reset(); o.mk_raygen(); o.mk_next_vertex(); o.mk_sample_bsdf(); o.mk_next_vertex(); o.mk_sample_bsdf(); o.mk_splat_preview(); o.postprocess()
    assert np.array_equal(t.read_pixels(0).view(np.uint32), o.read_pixels(0).view(np.uint32))
    for _ in range(7):
        t.update()
        o.mk_raygen(); o.mk_next_vertex(); o.mk_sample_bsdf(); o.mk_splat(); o.postprocess()
    assert np.array_equal(t.read_pixels(0).view(np.uint32), o.read_pixels(0).view(np.uint32))
    assert np.array_equal(t.read_pixels(1).view(np.uint32), o.read_pixels(1).view(np.uint32))
    rows = t.run_benchmark(0.0, iterations=10).strip().split("\n")
    vals = [float(x) for x in rows[-1].split(";")[1:]]
    assert vals[1] > 0 and vals[2] > 0 and vals[5] > 0                 # MK: primary, extension and samples come from the device counters
    t.toggle_renderer()
    assert t.uses_wavefront
    with pytest.raises(RuntimeError, match="numTasks"):
        t2 = Tracer(w, h, 0, 1000); t2.init(w, h, "proc:kitchen:2000:7"); t2.render_single(1)


@pytest.mark.gpu
def test_cpp_tracer_reference_format_caches(tmp_path):
    """Tracer::initHierarchy / saveState / loadState with the reference's file names and layouts (src/tracer.cpp:573-590, 625-684):
    <dir>/hierarchy_<XXH64 of the scene file>.bin and <dir>/state_<hash>.dat (44 four-byte items)."""
    import xxhash
    from fluctus_amd.tracer import Tracer
    obj = tmp_path / "quad.obj"
    obj.write_text("v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nv -1 1 -1\nv 1 1 -1\nv 1 2 1\nf 1 2 3\nf 1 3 4\nf 5 6 7\n")
    hdir, sdir = tmp_path / "hierarchies", tmp_path / "states"
    hdir.mkdir(); sdir.mkdir()
    w, h = 48, 32
    t = Tracer(w, h, 0, 4096)
    t.set_cache_dirs(str(hdir), str(sdir))
    t.init(w, h, str(obj))
    want = str(xxhash.xxh64(obj.read_bytes(), seed=0).intdigest())
    assert t.scene_hash == want
    cache = hdir / f"hierarchy_{want}.bin"
    assert cache.exists()
    nodes, idx = host.bvh_import(str(cache))
    assert idx.size >= 3 and nodes.size >= 1
    p = t.params
    wire.look_at(p, (0.0, 3.0, 4.0), (0.0, 0.5, 0.0))
    p["maxBounces"], p["envMapStrength"], p["exposure"], p["tmOperator"], p["useRoulette"] = 5, 2.5, 1.75, 1, 1
    t.params = p
    for _ in range(4):
        t.update()
    img = t.read_pixels(0)
    assert t.save_state()
    sfile = sdir / f"state_{want}.dat"
    assert sfile.stat().st_size == 44 * 4
    raw = np.frombuffer(sfile.read_bytes(), "<f4")
    assert np.isclose(raw[3], float(p["camera"]["fov"])) and np.allclose(raw[9:12], [p["camera"]["pos"][k] for k in "xyz"])   # after rotation(2), speed, fov, focalDist, aperture, dir(3)
    # a second tracer picks up both caches: same hierarchy (same image), same parameters
    t2 = Tracer(w, h, 0, 4096)
    t2.set_cache_dirs(str(hdir), str(sdir))
    t2.init(w, h, str(obj))
    assert t2.load_state()
    q = t2.params
    for f in ("maxBounces", "envMapStrength", "exposure", "tmOperator", "useRoulette", "useAreaLight", "useEnvMap", "sampleExpl", "sampleImpl"):
        assert q[f] == p[f], f
    for f in ("pos", "dir", "up", "right"):
        assert all(q["camera"][f][k] == p["camera"][f][k] for k in "xyz")
    for _ in range(4):
        t2.update()
    assert common.fb_close(t2.read_pixels(0), img)
    t3 = Tracer(w, h, 0, 4096)
    t3.set_cache_dirs(str(hdir), str(tmp_path / "nowhere"))
    t3.init(w, h, str(obj))
    assert not t3.load_state()


@pytest.mark.gpu
def test_cpp_tracer_drives_several_ranks_from_one_process():
    """Tracer(width, height, devices=[0, 0, 0]): the C++ host owns the multi-GPU path -- three contexts (here on the
    one device of the box; on a node: one per GPU + an RCCL communicator), scene replicated, pixel-interleaved partition, every
    enqueue fanned out, per-rank counters and cursors, tiles gathered and de-interleaved natively (flx_gather_local).
    Each rank is compared with an oracle context holding the same partition: counters exact, its pixels of the gathered image
    identical in sample counts and within the float-atomic tolerance in the sums."""
    from fluctus_amd.tracer import Tracer
    from fluctus_amd import multi
    from oracle.binding import OracleContext
    w, h, n, R = 90, 62, 4096, 3
    scene = "proc:conference:12000:43"
    t = Tracer(w, h, [0] * R, n)
    assert t.num_ranks == R
    t.set_option("extend_tree", 2)           # every rank: the reference's visit order, so that the comparison below is exact
    t.init(w, h, scene)
    p = t.params
    wire.look_at(p, (0.0, 1.2, 2.6), (0.0, 0.2, 0.0))
    p["maxBounces"], p["wfSeparateQueues"] = 5, 1
    t.params = p
    p = t.params
    d = host.generate_scene("conference", 12000, 43)
    host.build_bvh(d, "sbvh")
    orc = []
    for r in range(R):
        o = OracleContext(n, threads=8)
        o.upload_scene(d); o.set_partition(r, R); o.set_params(p)
        orc.append(o)
    tot = t.update()
    want = sum(driver.first_frame(o, p, multi.local_pixel_count(w * h, r, R)) for r, o in enumerate(orc))
    assert (tot == want).all(), (tot, want)
    for _ in range(6):
        tot = t.update()
        want = sum(driver.benchmark_iteration(o, multi.local_pixel_count(w * h, r, R)) for r, o in enumerate(orc))
        assert (tot == want).all(), (tot, want)
    full = t.read_accumulation()
    assert full.shape == (w * h, 4)
    for r, o in enumerate(orc):
        lp = multi.local_pixel_count(w * h, r, R)
        po = o.read_pixels(0)[:lp]
        assert np.array_equal(full[r::R][:, 3], po[:, 3]), f"rank {r}: sample counts"
        assert common.fb_close(full[r::R], po), f"rank {r}: radiance sums"
    csv = t.run_benchmark(0.0, iterations=6)
    assert len(csv.strip().split("\n")) >= 2
    with pytest.raises(RuntimeError, match="single-GPU"):
        t.render_single(1)
