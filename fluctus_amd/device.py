"""ctypes binding of libfluctus_hip.so (include/fluctus_hip.h).

HipContext mirrors the reference's CLContext method for method (reference: src/clcontext.hpp:31-79):
uploadSceneData -> upload_scene, createEnvMap -> upload_envmap, updateParams -> set_params,
enqueueWf*Kernel -> wf_*, enqueueClearWfQueues -> clear_queues, enqueueGetCounters -> get_counters,
finishQueue -> finish, updatePixelIndex/resetPixelIndex -> pixel_index_update/reset.
There is NO CPU fallback: without the library or without a GPU every call raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

# every symbol include/fluctus_hip.h declares (checked by tests/test_abi.py against the header)
SYMBOLS = ["flx_create", "flx_destroy", "flx_last_error", "flx_upload_scene", "flx_upload_envmap", "flx_set_params",
           "flx_wf_reset", "flx_wf_raygen", "flx_wf_extend", "flx_wf_shadow", "flx_wf_logic", "flx_wf_materials",
           "flx_clear_queues", "flx_get_counters_async", "flx_finish", "flx_pixel_index_update", "flx_pixel_index_reset",
           "flx_end_iteration_async", "flx_counter_totals", "flx_num_tasks", "flx_postprocess", "flx_read_pixels", "flx_set_partition", "flx_local_pixels",
           "flx_copy_pixels_to_device", "flx_stream", "flx_group_unique_id", "flx_group_init", "flx_group_init_local", "flx_gather", "flx_gather_local", "flx_group_destroy", "flx_group_info", "flx_profile_enable", "flx_profile_get", "flx_profile_reset",
           "flx_trace_stats_enable", "flx_trace_stats_get", "flx_trace_stats_get_ex", "flx_trace_stats_get_all", "flx_scene_info", "flx_trace_stats_reset", "flx_state_export", "flx_state_import", "flx_math_probe", "flx_env_sample_table",
           "flx_queue_read", "flx_queue_write", "flx_set_counters", "flx_set_option", "flx_get_option", "flx_mk_reset", "flx_mk_raygen", "flx_mk_next_vertex",
           "flx_mk_sample_bsdf", "flx_mk_splat", "flx_mk_splat_preview", "flx_mk_stats_async", "flx_mk_stats_reset"]

KERNELS = {"reset": 0, "raygen": 1, "extend": 2, "shadow": 3, "logic": 4, "materials": 5, "postprocess": 6, "trace_span": 7, "logic_fused": 8}


def _preload_torch_runtime():
    """PyTorch-ROCm bundles its own libamdhip64 / HSA runtime.  If libfluctus_hip.so (linked against /opt/rocm) is loaded
    first and torch later, the process ends up with two HSA runtimes and the first one no longer sees the GPU
    (scripts/order_probe.sh).  Importing torch first makes both share one runtime; without torch nothing is needed."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib_path():
    # FLX_HIP_LIB selects an A/B build of the same library (scripts/build_variants.py); default = the shipped one
    return os.environ.get("FLX_HIP_LIB") or os.path.join(_HERE, "libfluctus_hip.so")


def lib():
    global _lib
    if _lib is None:
        _preload_torch_runtime()
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing -- the HIP extension is required (no fallback); run __graft_entry__.build()")
        L = C.CDLL(path)
        for s in SYMBOLS:
            getattr(L, s)              # raises AttributeError if the library does not export it
        L.flx_last_error.restype = C.c_char_p
        L.flx_last_error.argtypes = [C.c_void_p]
        L.flx_num_tasks.restype = C.c_uint32
        L.flx_local_pixels.restype = C.c_uint32
        L.flx_stream.restype = C.c_void_p
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class HipContext:
    name = "mi355x"

    def __init__(self, num_tasks, device_index=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.num_tasks = int(num_tasks)
        rc = self.L.flx_create(int(device_index), C.c_uint32(num_tasks), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("flx_create failed: " + self.L.flx_last_error(None).decode())
        self.params = None
        self._cnt = []

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("libfluctus_hip: " + self.L.flx_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.flx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, d):
        self._chk(self.L.flx_upload_scene(self.h, _p(d.tris), C.c_size_t(d.tris.size), _p(d.indices), C.c_size_t(d.indices.size),
                                          _p(d.nodes), C.c_size_t(d.nodes.size), _p(d.materials), C.c_size_t(d.materials.size),
                                          _p(d.texdesc), C.c_size_t(d.texdesc.size), _p(d.texdata), C.c_size_t(d.texdata.size)))

    def upload_envmap(self, e):
        self._chk(self.L.flx_upload_envmap(self.h, _p(e.rgb), e.w, e.h, _p(e.prob), _p(e.alias), _p(e.pdf)))

    def set_params(self, p):
        self.params = p.copy()
        self._chk(self.L.flx_set_params(self.h, _p(np.ascontiguousarray(self.params).reshape(1))))

    def set_partition(self, rank, nranks):
        self._chk(self.L.flx_set_partition(self.h, C.c_uint32(rank), C.c_uint32(nranks)))

    def local_pixels(self):
        return int(self.L.flx_local_pixels(self.h))

    def wf_reset(self): self._chk(self.L.flx_wf_reset(self.h))
    def wf_raygen(self): self._chk(self.L.flx_wf_raygen(self.h))
    def wf_extend(self): self._chk(self.L.flx_wf_extend(self.h))
    def wf_shadow(self): self._chk(self.L.flx_wf_shadow(self.h))
    def wf_logic(self, first=False): self._chk(self.L.flx_wf_logic(self.h, int(bool(first))))
    def wf_materials(self): self._chk(self.L.flx_wf_materials(self.h))
    def postprocess(self): self._chk(self.L.flx_postprocess(self.h))
    def clear_queues(self): self._chk(self.L.flx_clear_queues(self.h))
    # microkernel integrator
    def mk_reset(self): self._chk(self.L.flx_mk_reset(self.h))
    def mk_raygen(self): self._chk(self.L.flx_mk_raygen(self.h))
    def mk_next_vertex(self): self._chk(self.L.flx_mk_next_vertex(self.h))
    def mk_sample_bsdf(self): self._chk(self.L.flx_mk_sample_bsdf(self.h))
    def mk_splat(self): self._chk(self.L.flx_mk_splat(self.h))
    def mk_splat_preview(self): self._chk(self.L.flx_mk_splat_preview(self.h))

    def mk_stats(self, reset=False):
        out = np.zeros(4, np.uint32)
        self._chk(self.L.flx_mk_stats_async(self.h, _p(out)))
        if reset:
            self._chk(self.L.flx_mk_stats_reset(self.h))
        self.finish()
        return out

    def get_counters(self):
        """Asynchronous: the returned array is filled by the next finish()."""
        out = np.zeros(8, np.uint32)
        self._cnt.append(out)           # keep alive until finish
        self._chk(self.L.flx_get_counters_async(self.h, _p(out)))
        return out

    def finish(self):
        self._chk(self.L.flx_finish(self.h))
        self._cnt.clear()

    def set_counters(self, c):
        c = np.ascontiguousarray(c, np.uint32)
        self._chk(self.L.flx_set_counters(self.h, _p(c)))

    def pixel_index_update(self, npix, nnew): self._chk(self.L.flx_pixel_index_update(self.h, C.c_uint32(npix), C.c_uint32(nnew)))
    def pixel_index_reset(self): self._chk(self.L.flx_pixel_index_reset(self.h))

    def end_iteration_async(self): self._chk(self.L.flx_end_iteration_async(self.h))

    def counter_totals(self, reset=False):
        out = np.zeros(8, np.uint64)
        self._chk(self.L.flx_counter_totals(self.h, _p(out), int(reset)))
        return out

    def read_pixels(self, which=0):
        out = np.zeros((self.local_pixels(), 4), np.float32)
        self._chk(self.L.flx_read_pixels(self.h, which, _p(out)))
        return out

    # multi-GPU group (RCCL); see include/fluctus_hip.h
    def group_init(self, rank, nranks, unique_id):
        """One process per GPU: ncclCommInitRank + the pixel partition.  unique_id: the 128 bytes of group_unique_id() made on rank 0."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self.L.flx_group_init(self.h, C.c_uint32(rank), C.c_uint32(nranks), buf))

    def group_info(self):
        """(ncclCommCount, ncclCommUserRank) as the communicator itself reports them."""
        out = (C.c_uint32 * 2)()
        self._chk(self.L.flx_group_info(self.h, out))
        return int(out[0]), int(out[1])

    def gather(self, root=0):
        """Collective over the group: returns the full (width*height, 4) accumulation image on `root`, None elsewhere."""
        npix = int(self.params["width"]) * int(self.params["height"])
        out = np.zeros((npix, 4), np.float32)
        self._chk(self.L.flx_gather(self.h, C.c_uint32(root), _p(out)))
        return out

    def copy_pixels_to_device(self, ptr):
        self._chk(self.L.flx_copy_pixels_to_device(self.h, C.c_void_p(ptr)))

    def state_export(self):
        out = np.zeros((64, self.num_tasks), np.float32)
        self._chk(self.L.flx_state_export(self.h, _p(out)))
        return out

    def env_sample_table(self, w, h):
        """(w * h, 8) float32: the per-texel light-sample table built at upload_envmap (test hook)."""
        out = np.zeros((int(w) * int(h), 8), np.float32)
        self._chk(self.L.flx_env_sample_table(self.h, _p(out)))
        return out

    def math_probe(self, fn, a, b):
        """include/flx_math.h function `fn` over operand arrays on the device; bit patterns (test hook)."""
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
        out = np.zeros(a.size, np.uint32)
        self._chk(self.L.flx_math_probe(self.h, int(fn), _p(a), _p(b), C.c_uint32(a.size), _p(out)))
        return out

    def state_import(self, st):
        st = np.ascontiguousarray(st, np.float32)
        assert st.shape == (64, self.num_tasks)
        self._chk(self.L.flx_state_import(self.h, _p(st)))

    def queue_read(self, q):
        out = np.zeros(self.num_tasks, np.uint32)
        self._chk(self.L.flx_queue_read(self.h, q, _p(out)))
        return out

    def queue_write(self, q, arr):
        arr = np.ascontiguousarray(arr, np.uint32)
        self._chk(self.L.flx_queue_write(self.h, q, _p(arr), C.c_uint32(arr.size)))

    # measurement
    def profile_enable(self, on=True): self._chk(self.L.flx_profile_enable(self.h, int(on)))
    def profile_reset(self): self._chk(self.L.flx_profile_reset(self.h))

    def profile_get(self):
        out = {}
        for name, k in KERNELS.items():
            ms, n = C.c_double(), C.c_uint64()
            self._chk(self.L.flx_profile_get(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def trace_stats_enable(self, on=True): self._chk(self.L.flx_trace_stats_enable(self.h, int(on)))
    def reset_stats(self): self._chk(self.L.flx_trace_stats_reset(self.h))

    def stats(self):
        out = np.zeros(7, np.uint64)
        self._chk(self.L.flx_trace_stats_get(self.h, _p(out)))
        return dict(ext_rays=int(out[0]), ext_inner=int(out[1]), ext_tri=int(out[2]), ext_hits=int(out[3]),
                    shadow_inner=int(out[4]), shadow_tri=int(out[5]), shadow_rays=int(out[6]))

    def leaf_stats(self):
        """Leaf visits of the extension / shadow traversal (wide kernels only)."""
        out = np.zeros(24, np.uint64)
        self._chk(self.L.flx_trace_stats_get_all(self.h, _p(out)))
        return dict(ext_leaf=int(out[16]), shadow_leaf=int(out[17]))

    def scene_info(self):
        out = np.zeros(8, np.uint32)
        self._chk(self.L.flx_scene_info(self.h, _p(out)))
        k = ("wide_nodes", "wide_leaf_f4", "wide_stack_bound", "nested", "binary_depth", "spill_levels", "binary_records", "max_leaf")
        return {n: int(v) for n, v in zip(k, out)}

    def wave_stats(self):
        """Wave-level trip counts of the traversal loops (see flx_trace_stats_get_ex)."""
        out = np.zeros(16, np.uint64)
        self._chk(self.L.flx_trace_stats_get_ex(self.h, _p(out)))
        k = ("outer", "inner", "leaf", "tri")
        return dict(ext={n: int(out[8 + i]) for i, n in enumerate(k)}, shadow={n: int(out[12 + i]) for i, n in enumerate(k)}, ext_max_inner_sum=int(out[7]))

    def set_option(self, name, value): self._chk(self.L.flx_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int()
        self._chk(self.L.flx_get_option(self.h, name.encode(), C.byref(v)))
        return v.value


def group_unique_id():
    """ncclGetUniqueId (rank 0 of a multi-process job); ship the bytes to the other ranks and pass them to HipContext.group_init."""
    buf = (C.c_char * 128)()
    if lib().flx_group_unique_id(buf) != 0:
        raise RuntimeError("flx_group_unique_id failed: " + lib().flx_last_error(None).decode())
    return bytes(buf)


def _handles(ctxs):
    return (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])


def group_init_local(ctxs):
    """Single process: contexts 0..n-1 become ranks 0..n-1 of one group (RCCL communicator if on distinct devices)."""
    if lib().flx_group_init_local(_handles(ctxs), C.c_uint32(len(ctxs))) != 0:
        raise RuntimeError("flx_group_init_local: " + lib().flx_last_error(ctxs[0].h).decode())


def gather_local(ctxs, root=0):
    p = ctxs[root].params
    out = np.zeros((int(p["width"]) * int(p["height"]), 4), np.float32)
    if lib().flx_gather_local(_handles(ctxs), C.c_uint32(len(ctxs)), C.c_uint32(root), _p(out)) != 0:
        raise RuntimeError("flx_gather_local: " + lib().flx_last_error(ctxs[root].h).decode())
    return out
