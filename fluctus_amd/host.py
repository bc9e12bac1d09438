"""ctypes binding of libfluctus_host.so (C++ scene / BVH / env-map preparation)."""
import ctypes as C
import os
import numpy as np
from .wire import TRIANGLE, NODE, MATERIAL, TEXDESC

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        try:                              # see device._preload_torch_runtime: the C++ HipContext dlopens libfluctus_hip.so later
            import torch  # noqa: F401
        except Exception:
            pass
        path = os.path.join(_HERE, "libfluctus_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing -- run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(path)
        _lib.fh_last_error.restype = C.c_char_p
    return _lib


def _chk(rc):
    if rc != 0:
        raise RuntimeError("libfluctus_host: " + lib().fh_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class SceneData:
    """Triangles/materials/textures in wire format + BVH arrays."""

    def __init__(self):
        self.tris = self.materials = self.texdesc = self.texdata = None
        self.nodes = self.indices = None
        self.world_radius = 1.0
        self.type_bits = 0
        self.world_up = (0.0, 1.0, 0.0)
        self.bvh_metrics = None


def _scene_to_data(h):
    L = lib()
    nt, nm, nx, tb, bits = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
    _chk(L.fh_scene_counts(h, C.byref(nt), C.byref(nm), C.byref(nx), C.byref(tb), C.byref(bits)))
    d = SceneData()
    d.tris = np.zeros(nt.value, TRIANGLE)
    d.materials = np.zeros(nm.value, MATERIAL)
    d.texdesc = np.zeros(nx.value, TEXDESC)
    d.texdata = np.zeros(tb.value, np.uint8)
    _chk(L.fh_scene_get(h, _p(d.tris), _p(d.materials), _p(d.texdesc), _p(d.texdata)))
    d.type_bits = bits.value
    up = np.zeros(3, np.float32)
    L.fh_scene_world_up(h, up.ctypes.data_as(C.c_void_p))
    d.world_up = tuple(float(x) for x in up)
    return d


def load_scene(path):
    L = lib()
    h = C.c_void_p()
    _chk(L.fh_scene_create(C.byref(h)))
    try:
        _chk(L.fh_scene_load(h, path.encode()))
        return _scene_to_data(h)
    finally:
        L.fh_scene_destroy(h)


def generate_scene(kind, target_tris, seed):
    L = lib()
    h = C.c_void_p()
    _chk(L.fh_scene_create(C.byref(h)))
    try:
        _chk(L.fh_scene_generate(h, kind.encode(), C.c_uint32(target_tris), C.c_uint32(seed)))
        return _scene_to_data(h)
    finally:
        L.fh_scene_destroy(h)


def load_png(path):
    """RGBA8 texture, lower-left origin (row 0 = bottom scanline), as the reference's DevIL setup delivers it."""
    L = lib()
    w, h = C.c_uint32(), C.c_uint32()
    _chk(L.fh_png_load(path.encode(), C.byref(w), C.byref(h), None))
    out = np.zeros((h.value, w.value, 4), np.uint8)
    _chk(L.fh_png_load(path.encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p)))
    return out


load_texture = load_png      # PNG or JPEG, decided by the file signature


def decode_jpeg(data):
    """JPEG bytes -> (h, w, 3) uint8, top-left origin: the bytes libjpeg's default decode path produces."""
    L = lib()
    buf = np.frombuffer(bytes(data), np.uint8)
    w, h = C.c_uint32(), C.c_uint32()
    _chk(L.fh_jpeg_decode(buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size), C.byref(w), C.byref(h), None, C.c_uint64(0)))
    out = np.zeros((h.value, w.value, 3), np.uint8)
    _chk(L.fh_jpeg_decode(buf.ctypes.data_as(C.c_void_p), C.c_uint64(buf.size), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p), C.c_uint64(out.size)))
    return out


BVH_MODES = {"sbvh": 0, "sah": 1, "binned": 2}


def build_bvh(d, mode="sbvh", threads=0, job_size=0):
    """threads: 0 = all usable cores (the SBVH builder is parallel; the tree does not depend on the thread count), 1 = serial."""
    L = lib()
    h = C.c_void_p()
    _chk(L.fh_bvh_build_ex(_p(d.tris), C.c_uint64(d.tris.size), BVH_MODES[mode], int(threads), C.c_uint64(job_size), C.byref(h)))
    try:
        nn, ni = C.c_uint64(), C.c_uint64()
        met = (C.c_uint32 * 4)()
        _chk(L.fh_bvh_counts(h, C.byref(nn), C.byref(ni), met))
        d.nodes = np.zeros(nn.value, NODE)
        d.indices = np.zeros(ni.value, np.uint32)
        wr = C.c_float()
        _chk(L.fh_bvh_get(h, _p(d.nodes), _p(d.indices), C.byref(wr)))
        d.world_radius = wr.value
        d.bvh_metrics = dict(depth=met[0], splits=met[1], duplicates=met[2], spatial_splits=met[3])
    finally:
        L.fh_bvh_destroy(h)
    return d


def wide_tree_check(d):
    """Build the traversal kernels' 4-wide quantised tree (csrc/flx_wide.h -- what flx_upload_scene builds) on the CPU and check its
    invariants: conservative quantised boxes, every binary leaf exactly once with its box / count / triangle order, every wide node
    reachable once, unused slots inverted.  Raises on a violation; returns the tree's figures."""
    out = (C.c_uint64 * 8)()
    areas = (C.c_double * 2)()
    _chk(lib().fh_wide_tree_areas(_p(d.nodes), C.c_uint64(d.nodes.size), _p(d.tris), C.c_uint64(d.tris.size),
                                  _p(d.indices), C.c_uint64(d.indices.size), out, areas))
    return dict(area_exact=areas[0], area_quantised=areas[1], wide_nodes=out[0], leaf_float4s=out[1], max_stack=out[2], nested=bool(out[3]), leaves=out[4],
                slots_used={2: out[5], 3: out[6], 4: out[7]})


def bvh_export(d, path, mode="sbvh"):
    """Build and write the hierarchy cache file (the reference's on-disk format, host/bvh.hpp)."""
    L = lib()
    h = C.c_void_p()
    _chk(L.fh_bvh_build(_p(d.tris), C.c_uint64(d.tris.size), BVH_MODES[mode], C.byref(h)))
    try:
        _chk(L.fh_bvh_export(h, path.encode()))
    finally:
        L.fh_bvh_destroy(h)


def bvh_import(path):
    """(nodes, indices) read back from a hierarchy cache file."""
    L = lib()
    h = C.c_void_p()
    _chk(L.fh_bvh_import(path.encode(), C.byref(h)))
    try:
        nn, ni = C.c_uint64(), C.c_uint64()
        met = (C.c_uint32 * 4)()
        _chk(L.fh_bvh_counts(h, C.byref(nn), C.byref(ni), met))
        nodes, idx = np.zeros(nn.value, NODE), np.zeros(ni.value, np.uint32)
        wr = C.c_float()
        _chk(L.fh_bvh_get(h, _p(nodes), _p(idx), C.byref(wr)))
        bvh_import.world_radius = wr.value           # of the last import (root box), same arithmetic as a fresh build
        return nodes, idx
    finally:
        L.fh_bvh_destroy(h)


def bvh_export_arrays(d, path):
    """Write d.nodes / d.indices as a hierarchy cache file without rebuilding (same bytes as BVH::exportTo)."""
    rec = np.zeros(d.nodes.size, np.dtype([("box", "<f4", 6), ("istart", "<u4"), ("parent", "<i4"), ("nprims", "u1")]))
    for k, ax in enumerate("xyz"):
        rec["box"][:, k] = d.nodes["bmin"][ax]
        rec["box"][:, 3 + k] = d.nodes["bmax"][ax]
    rec["istart"], rec["parent"], rec["nprims"] = d.nodes["iStartOrRight"], d.nodes["parent"], d.nodes["nPrims"]
    with open(path, "wb") as f:
        f.write(np.uint32(d.indices.size).tobytes()); f.write(np.ascontiguousarray(d.indices, "<u4").tobytes())
        f.write(np.uint32(d.nodes.size).tobytes()); f.write(rec.tobytes())


def xxh64(data, seed=0):
    L = lib()
    L.fh_xxh64.restype = C.c_uint64
    buf = np.frombuffer(bytes(data), np.uint8)
    return int(L.fh_xxh64(buf.ctypes.data_as(C.c_void_p) if buf.size else None, C.c_uint64(buf.size), C.c_uint64(seed)))


class EnvMap:
    def __init__(self, w, h, rgb, prob, alias, pdf):
        self.w, self.h, self.rgb, self.prob, self.alias, self.pdf = w, h, rgb, prob, alias, pdf


def _env_to_py(h):
    L = lib()
    w, hh = C.c_int(), C.c_int()
    L.fh_envmap_dims(h, C.byref(w), C.byref(hh))
    n = w.value * hh.value
    rgb = np.zeros(n * 3, np.float32)
    prob = np.zeros(n, np.float32)
    alias = np.zeros(n, np.int32)
    pdf = np.zeros(n, np.float32)
    _chk(L.fh_envmap_get(h, _p(rgb), _p(prob), _p(alias), _p(pdf)))
    return EnvMap(w.value, hh.value, rgb, prob, alias, pdf)


def load_envmap(path):
    L = lib()
    h = C.c_void_p()
    _chk(L.fh_envmap_load(path.encode(), C.byref(h)))
    try:
        return _env_to_py(h)
    finally:
        L.fh_envmap_destroy(h)


def envmap_from_rgb(w, h, rgb):
    L = lib()
    hd = C.c_void_p()
    rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1)
    _chk(L.fh_envmap_from_memory(w, h, _p(rgb), C.byref(hd)))
    try:
        return _env_to_py(hd)
    finally:
        L.fh_envmap_destroy(hd)


def synthetic_sky(w=512, h=256, seed=7):
    """Deterministic HDR sky (gradient + a few bright lobes): the env map that travels to the GPU
    box, where /root/reference (assets/env_maps/night.hdr) does not exist."""
    v = (np.arange(h, dtype=np.float32) + 0.5) / h
    u = (np.arange(w, dtype=np.float32) + 0.5) / w
    uu, vv = np.meshgrid(u, v)
    up = np.clip(1.0 - 2.0 * vv, 0.0, 1.0)
    img = np.stack([0.25 + 0.35 * up, 0.35 + 0.45 * up, 0.55 + 0.75 * up], -1).astype(np.float32)
    img[vv > 0.5] *= 0.25
    rng = np.random.RandomState(seed)
    for _ in range(4):
        cu, cv, s, a = rng.uniform(0, 1), rng.uniform(0.1, 0.4), rng.uniform(0.01, 0.04), rng.uniform(20, 200)
        du = np.minimum(np.abs(uu - cu), 1.0 - np.abs(uu - cu))
        img += (a * np.exp(-((du ** 2) + (vv - cv) ** 2) / (2 * s * s)))[..., None].astype(np.float32) * \
            np.array([1.0, 0.9, 0.7], np.float32)
    return envmap_from_rgb(w, h, img.astype(np.float32))
