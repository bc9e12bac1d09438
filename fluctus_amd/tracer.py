"""ctypes binding of the C++ headless Tracer (fluctus_amd/host/tracer.cpp) in libfluctus_host.so."""
import ctypes as C
import numpy as np
from . import host
from .wire import RENDER_PARAMS


class Tracer:
    def __init__(self, width, height, device=0, num_tasks=1 << 20):
        """device: one HIP device index, or a list of them (one process driving several GPUs: devices[0] is the root; the same
        index may repeat -- a 1-GPU stand-in for N ranks)."""
        self.L = host.lib()
        self.h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            devs = (C.c_int * len(device))(*[int(d) for d in device])
            host._chk(self.L.fh_tracer_create_multi(int(width), int(height), devs, len(device), C.c_uint32(num_tasks), C.byref(self.h)))
        else:
            host._chk(self.L.fh_tracer_create(int(width), int(height), int(device), C.c_uint32(num_tasks), C.byref(self.h)))

    @property
    def num_ranks(self):
        return int(self.L.fh_tracer_num_ranks(self.h))

    def read_accumulation(self):
        """Full-resolution accumulation image (rgb sum, sample count), gathered from all ranks."""
        p = self.params
        out = np.zeros((int(p["width"]) * int(p["height"]), 4), np.float32)
        host._chk(self.L.fh_tracer_read_accumulation(self.h, out.ctypes.data_as(C.c_void_p), C.c_uint64(out.size)))
        return out

    def close(self):
        if self.h:
            self.L.fh_tracer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init(self, width, height, scene):
        host._chk(self.L.fh_tracer_init(self.h, int(width), int(height), scene.encode()))

    def set_envmap(self, path):
        host._chk(self.L.fh_tracer_set_envmap(self.h, path.encode()))

    @property
    def params(self):
        p = np.zeros(1, RENDER_PARAMS)
        host._chk(self.L.fh_tracer_params(self.h, p.ctypes.data_as(C.c_void_p), None))
        return p.reshape(())

    @params.setter
    def params(self, p):
        a = np.ascontiguousarray(p).reshape(1)
        host._chk(self.L.fh_tracer_params(self.h, None, a.ctypes.data_as(C.c_void_p)))

    def update(self):
        cnt = np.zeros(8, np.uint32)
        host._chk(self.L.fh_tracer_update(self.h, cnt.ctypes.data_as(C.c_void_p)))
        return cnt

    def render_single(self, spp, denoise=False):
        """Tracer::renderSingle (src/tracer.cpp:95-187): exactly spp samples per pixel on the microkernel integrator;
        denoise=True also fills the denoiser feature buffers (read_pixels(2) albedo, read_pixels(3) normals)."""
        host._chk(self.L.fh_tracer_render_single(self.h, int(spp), int(bool(denoise))))

    def set_denoiser(self, on):
        host._chk(self.L.fh_tracer_set_denoiser(self.h, int(bool(on))))

    def set_option(self, name, value):
        """HipContext::setOption -> flx_set_option (e.g. "extend_tree", 2 for the reference's bit-exact visit order)."""
        host._chk(self.L.fh_tracer_set_option(self.h, name.encode(), int(value)))

    def toggle_renderer(self):
        host._chk(self.L.fh_tracer_toggle_renderer(self.h))

    @property
    def uses_wavefront(self):
        return bool(self.L.fh_tracer_uses_wavefront(self.h))

    def stats(self):
        """RenderStats accumulated on the host: primary, extension, shadow rays, samples."""
        out = np.zeros(4, np.uint64)
        host._chk(self.L.fh_tracer_stats(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_cache_dirs(self, hierarchies="", states=""):
        """Directories of the reference-format caches: hierarchy_<hash>.bin (BVH) and state_<hash>.dat (camera/light/sampling state)."""
        self.L.fh_tracer_set_cache_dirs(self.h, hierarchies.encode(), states.encode())

    def save_state(self):
        return self.L.fh_tracer_save_state(self.h) == 0

    def load_state(self):
        return self.L.fh_tracer_load_state(self.h) == 0

    @property
    def scene_hash(self):
        buf = C.create_string_buffer(64)
        host._chk(self.L.fh_tracer_scene_hash(self.h, buf, C.c_uint64(64)))
        return buf.value.decode()

    def run_benchmark(self, seconds=1.0, iterations=0):
        buf = C.create_string_buffer(1 << 20)
        host._chk(self.L.fh_tracer_run_benchmark(self.h, C.c_double(seconds), int(iterations), buf, C.c_uint64(len(buf))))
        return buf.value.decode()

    def read_pixels(self, which=0):
        p = self.params
        out = np.zeros((int(p["width"]) * int(p["height"]), 4), np.float32)
        host._chk(self.L.fh_tracer_read_pixels(self.h, which, out.ctypes.data_as(C.c_void_p), C.c_uint64(out.size)))
        return out

    def save_image(self, path):
        host._chk(self.L.fh_tracer_save_image(self.h, path.encode()))
