"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The interface mirrors fluctus_amd.device.HipContext method for method so one driver loop runs both.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing -- build it with `make -C oracle`")
        _lib = C.CDLL(path)
        _lib.orc_math.restype = C.c_float
        _lib.orc_math.argtypes = [C.c_int, C.c_float, C.c_float]
        _lib.orc_hash.restype = C.c_uint32
        _lib.orc_hash.argtypes = [C.c_uint32]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def math_array(fn, a, b):
    """include/flx_math.h function `fn` (see orc_math) over operand arrays on the host; bit patterns."""
    import numpy as np
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.zeros(a.size, np.uint32)
    lib().orc_math_array(int(fn), _p(a), _p(b), C.c_uint32(a.size), _p(out))
    return out


class _Prefixed:
    """Attribute access L.orc_xyz -> getattr(lib, prefix + 'xyz') so one class serves both libraries."""

    def __init__(self, cdll, prefix):
        self._l, self._p = cdll, prefix

    def __getattr__(self, name):
        assert name.startswith("orc_")
        return getattr(self._l, self._p + name[4:])


_ref_lib = None


def ref_lib():
    """oracle/_ref/libfluctus_ref.so: the reference's own kernels compiled for x86-64 (this container only)."""
    global _ref_lib
    if _ref_lib is None:
        path = os.path.join(_HERE, "_ref", "libfluctus_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        _ref_lib = C.CDLL(path)
    return _ref_lib


_refbvh_lib = None


def refbvh_lib():
    """oracle/_ref/libfluctus_refbvh.so: the reference's own object-split BVH builder (src/bvh.cpp, src/bvhnode.cpp) compiled
    unmodified (oracle/ref/Makefile, oracle/ref/bvh_driver.cpp); this container only."""
    global _refbvh_lib
    if _refbvh_lib is None:
        path = os.path.join(_HERE, "_ref", "libfluctus_refbvh.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        _refbvh_lib = C.CDLL(path)
    return _refbvh_lib


def refbvh_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libfluctus_refbvh.so"))


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libfluctus_ref.so"))


class OracleContext:
    name = "oracle"

    def __init__(self, num_tasks, threads=1, _ref=False):
        self.L = _Prefixed(ref_lib(), "ref_") if _ref else lib()
        self.h = C.c_void_p()
        self.num_tasks = int(num_tasks)
        assert self.L.orc_create(C.c_uint32(num_tasks), C.byref(self.h)) == 0
        self.L.orc_set_threads(self.h, int(threads))
        self.params = None

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def upload_scene(self, d):
        self.L.orc_upload_scene(self.h, _p(d.tris), C.c_size_t(d.tris.size), _p(d.indices), C.c_size_t(d.indices.size),
                                _p(d.nodes), C.c_size_t(d.nodes.size), _p(d.materials), C.c_size_t(d.materials.size),
                                _p(d.texdesc), C.c_size_t(d.texdesc.size), _p(d.texdata), C.c_size_t(d.texdata.size))

    def upload_envmap(self, e):
        self.L.orc_upload_envmap(self.h, _p(e.rgb), e.w, e.h, _p(e.prob), _p(e.alias), _p(e.pdf))
        self._env_wh = (e.w, e.h)

    def env_sample_table(self):
        """(w * h, 8) float32: what next-event estimation derives from every texel of the uploaded map (orc_env_sample_table)."""
        n = int(self._env_wh[0]) * int(self._env_wh[1])
        out = np.zeros((n, 8), np.float32)
        self.L.orc_env_sample_table(self.h, _p(out))
        return out

    def set_params(self, p):
        self.params = p.copy()
        self.L.orc_set_params(self.h, _p(np.ascontiguousarray(self.params).reshape(1)))

    def set_partition(self, rank, nranks):
        self.L.orc_set_partition(self.h, C.c_uint32(rank), C.c_uint32(nranks))

    def set_option(self, name, value):
        """Only "denoiser" (the reference's USE_OPTIX_DENOISER build of the kernels); device tuning knobs do not exist here."""
        assert self.L.orc_set_option(self.h, name.encode(), int(value)) == 0, name

    def wf_reset(self): self.L.orc_wf_reset(self.h)
    def wf_raygen(self): self.L.orc_wf_raygen(self.h)
    def wf_extend(self): self.L.orc_wf_extend(self.h)
    def wf_shadow(self): self.L.orc_wf_shadow(self.h)
    def wf_logic(self, first=False): self.L.orc_wf_logic(self.h, int(bool(first)))
    def wf_materials(self): self.L.orc_wf_materials(self.h)
    def postprocess(self): self.L.orc_postprocess(self.h)
    # microkernel integrator
    def mk_reset(self): self.L.orc_mk_reset(self.h)
    def mk_raygen(self): self.L.orc_mk_raygen(self.h)
    def mk_next_vertex(self): self.L.orc_mk_next_vertex(self.h)
    def mk_sample_bsdf(self): self.L.orc_mk_sample_bsdf(self.h)
    def mk_splat(self): self.L.orc_mk_splat(self.h)
    def mk_splat_preview(self): self.L.orc_mk_splat_preview(self.h)

    def mk_stats(self, reset=False):
        out = np.zeros(4, np.uint32)
        self.L.orc_mk_stats(self.h, _p(out), int(reset))
        return out
    def clear_queues(self): self.L.orc_clear_queues(self.h)
    def finish(self): pass

    def get_counters(self):
        out = np.zeros(8, np.uint32)
        self.L.orc_get_counters(self.h, _p(out))
        return out

    def set_counters(self, c):
        c = np.ascontiguousarray(c, np.uint32)
        self.L.orc_set_counters(self.h, _p(c))

    def pixel_index_update(self, npix, nnew): self.L.orc_pixel_index_update(self.h, C.c_uint32(npix), C.c_uint32(nnew))
    def pixel_index_reset(self): self.L.orc_pixel_index_reset(self.h)

    def read_pixels(self, which=0):
        n = int(self.params["width"]) * int(self.params["height"])
        out = np.zeros((n, 4), np.float32)
        self.L.orc_read_pixels(self.h, which, _p(out))
        return out

    def state_export(self):
        out = np.zeros((64, self.num_tasks), np.float32)
        self.L.orc_state_export(self.h, _p(out))
        return out

    def state_import(self, st):
        st = np.ascontiguousarray(st, np.float32)
        assert st.shape == (64, self.num_tasks)
        self.L.orc_state_import(self.h, _p(st))

    def queue_read(self, q):
        out = np.zeros(self.num_tasks, np.uint32)
        self.L.orc_queue_read(self.h, q, _p(out))
        return out

    def queue_write(self, q, arr):
        arr = np.ascontiguousarray(arr, np.uint32)
        self.L.orc_queue_write(self.h, q, _p(arr), C.c_uint32(arr.size))

    def stats(self):
        out = np.zeros(7, np.uint64)
        self.L.orc_get_stats(self.h, _p(out))
        return dict(ext_rays=int(out[0]), ext_inner=int(out[1]), ext_tri=int(out[2]), ext_hits=int(out[3]),
                    shadow_inner=int(out[4]), shadow_tri=int(out[5]), shadow_rays=int(out[6]))

    def reset_stats(self): self.L.orc_reset_stats(self.h)


class RefContext(OracleContext):
    """Same interface, backed by the reference's own kernels (oracle/_ref)."""
    name = "reference"

    def __init__(self, num_tasks, threads=1):
        super().__init__(num_tasks, threads=threads, _ref=True)
