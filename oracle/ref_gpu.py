"""The reference's OWN wavefront kernels on the MI355X ("Pin 5").

TEST INFRASTRUCTURE: imported only by tests/ and bench.py's reference-timing leg; nothing in the product touches it.

oracle/_ref/gfx950/{ieee,fast}/*.co are /root/reference/src/wf_*.cl compiled UNMODIFIED for gfx950 by the committed recipe
(oracle/ref/Makefile, target `gfx950`) and linked with AMD's own OpenCL built-in library (opencl.bc / ocml.bc / ockl.bc) --
no builder-written stand-in for any built-in, no sequential NDRange driver: the code objects are loaded by ROCm's OpenCL
runtime on the GPU box (clCreateProgramWithBinary) and enqueued with the NDRanges of reference src/clcontext.cpp:765-850 and
the argument order of src/wf_*.cl (= the setArg names of src/kernel_impl.hpp).  The code objects are built in the build
container (the only place /root/reference exists) and travel to the GPU box like every other built artefact.

RefGpuContext mirrors oracle.binding.OracleContext / fluctus_amd.device.HipContext method for method, so the lockstep
helpers of the parity tests drive it unchanged.

Two loaders ("backends"):
  * "opencl": libOpenCL.so.1 (ICD loader -> ROCm's runtime).  The real thing for every kernel without an image argument.
  * "hip":    libamdhip64.so hipModuleLoadData / hipModuleLaunchKernel on the SAME code objects.  Needed for `logic`: its
              signature carries `read_only image2d_t envMap` (src/wf_logic.cl:31) even in the variants that never sample it,
              gfx950 has no image support (CL_DEVICE_IMAGE_SUPPORT = 0), so an OpenCL host cannot create the object
              clSetKernelArg wants; through the module API the (unused) descriptor pointer is passed as null.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NUM_COLS = 64
Q_RAYGEN, Q_EXTENSION, Q_SHADOW, Q_DIFFUSE, Q_GLOSSY, Q_GGX_REFL, Q_GGX_REFR, Q_DELTA = range(8)


class RefGpuUnavailable(RuntimeError):
    pass


def co_dir(flavour="ieee"):
    return os.path.join(_HERE, "_ref", "gfx950", flavour)


def available(flavour="ieee"):
    return os.path.exists(os.path.join(co_dir(flavour), "traceExtension.co"))


def available_env(flavour="ieee"):
    """the USE_ENV_MAP variants of `logic`, built with the image stand-in (oracle/ref/gfx950_image_standin.cl)"""
    return os.path.exists(os.path.join(co_dir(flavour), "logic_v14_imgstandin.co"))


# ------------------------------------------------------------------------------------------------ OpenCL backend
class _OpenCL:
    name = "opencl"

    def __init__(self):
        # Both runtimes in one process (the tests drive the HIP product beside this): HIP must be initialised BEFORE ROCm's OpenCL runtime --
        # the other way round hipGetDeviceCount reports no device afterwards (seen on the MI355X box, round 5).
        try:
            hip = C.CDLL("libamdhip64.so")
            n = C.c_int()
            hip.hipInit(0); hip.hipGetDeviceCount(C.byref(n))
        except OSError:
            pass
        try:
            L = C.CDLL("libOpenCL.so.1")
        except OSError as e:
            raise RefGpuUnavailable(f"libOpenCL.so.1 not loadable: {e}")
        self.L = L
        vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
        L.clCreateContext.restype = vp
        L.clCreateContext.argtypes = [vp, u32, C.POINTER(vp), vp, vp, C.POINTER(C.c_int)]
        L.clCreateCommandQueue.restype = vp
        L.clCreateCommandQueue.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_int)]
        L.clCreateProgramWithBinary.restype = vp
        L.clCreateProgramWithBinary.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.clBuildProgram.argtypes = [vp, u32, C.POINTER(vp), C.c_char_p, vp, vp]
        L.clGetProgramBuildInfo.argtypes = [vp, vp, u32, sz, vp, C.POINTER(sz)]
        L.clCreateKernel.restype = vp
        L.clCreateKernel.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
        L.clCreateBuffer.restype = vp
        L.clCreateBuffer.argtypes = [vp, C.c_uint64, sz, vp, C.POINTER(C.c_int)]
        L.clSetKernelArg.argtypes = [vp, u32, sz, vp]
        L.clEnqueueNDRangeKernel.argtypes = [vp, vp, u32, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), u32, vp, C.POINTER(vp)]
        L.clEnqueueReadBuffer.argtypes = [vp, vp, u32, sz, sz, vp, u32, vp, vp]
        L.clEnqueueWriteBuffer.argtypes = [vp, vp, u32, sz, sz, vp, u32, vp, vp]
        L.clFinish.argtypes = [vp]
        L.clReleaseMemObject.argtypes = [vp]
        L.clReleaseEvent.argtypes = [vp]
        L.clWaitForEvents.argtypes = [u32, C.POINTER(vp)]
        L.clGetEventProfilingInfo.argtypes = [vp, u32, sz, vp, C.POINTER(sz)]
        L.clGetDeviceInfo.argtypes = [vp, u32, sz, vp, C.POINTER(sz)]
        L.clGetPlatformIDs.argtypes = [u32, C.POINTER(vp), C.POINTER(u32)]
        L.clGetDeviceIDs.argtypes = [vp, C.c_uint64, u32, C.POINTER(vp), C.POINTER(u32)]
        nplat = u32()
        if L.clGetPlatformIDs(0, None, C.byref(nplat)) != 0 or nplat.value == 0:
            raise RefGpuUnavailable("no OpenCL platform")
        plats = (vp * nplat.value)()
        L.clGetPlatformIDs(nplat.value, plats, None)
        self.dev = None
        for p in plats:
            nd = u32()
            if L.clGetDeviceIDs(p, 4, 0, None, C.byref(nd)) != 0 or nd.value == 0:     # CL_DEVICE_TYPE_GPU
                continue
            devs = (vp * nd.value)()
            L.clGetDeviceIDs(p, 4, nd.value, devs, None)
            self.dev = vp(devs[0])
            break
        if self.dev is None:
            raise RefGpuUnavailable("OpenCL platform present but it exposes no GPU device")
        err = C.c_int()
        self.ctx = L.clCreateContext(None, 1, C.byref(self.dev), None, None, C.byref(err))
        if err.value != 0:
            raise RefGpuUnavailable(f"clCreateContext failed: {err.value}")
        self.q = L.clCreateCommandQueue(self.ctx, self.dev, 2, C.byref(err))        # CL_QUEUE_PROFILING_ENABLE
        if err.value != 0:
            raise RefGpuUnavailable(f"clCreateCommandQueue failed: {err.value}")
        self._progs = {}

    def device_name(self):
        buf = C.create_string_buffer(256)
        self.L.clGetDeviceInfo(self.dev, 0x102B, 256, buf, None)
        return buf.value.decode(errors="replace")

    def load(self, path):
        if path in self._progs:
            return self._progs[path]
        data = open(path, "rb").read()
        err, status = C.c_int(), C.c_int()
        ln = C.c_size_t(len(data))
        bin_ = C.c_char_p(data)
        prog = self.L.clCreateProgramWithBinary(self.ctx, 1, C.byref(self.dev), C.byref(ln), C.byref(bin_), C.byref(status), C.byref(err))
        if err.value != 0 or status.value != 0:
            raise RefGpuUnavailable(f"clCreateProgramWithBinary({os.path.basename(path)}) failed: err {err.value}, binary status {status.value}")
        rc = self.L.clBuildProgram(prog, 1, C.byref(self.dev), b"", None, None)
        if rc != 0:
            n = C.c_size_t()
            self.L.clGetProgramBuildInfo(prog, self.dev, 0x1183, 0, None, C.byref(n))
            log = C.create_string_buffer(n.value + 1)
            self.L.clGetProgramBuildInfo(prog, self.dev, 0x1183, n.value, log, None)
            raise RefGpuUnavailable(f"clBuildProgram({os.path.basename(path)}) failed ({rc}): {log.value.decode(errors='replace')}")
        self._progs[path] = prog
        return prog

    def kernel(self, prog, name):
        err = C.c_int()
        k = self.L.clCreateKernel(prog, name.encode(), C.byref(err))
        if err.value != 0:
            raise RefGpuUnavailable(f"clCreateKernel({name}) failed: {err.value}")
        return k

    def alloc(self, nbytes):
        err = C.c_int()
        b = self.L.clCreateBuffer(self.ctx, 1, max(16, int(nbytes)), None, C.byref(err))
        assert err.value == 0, f"clCreateBuffer({nbytes}): {err.value}"
        return b

    def free(self, b):
        if b:
            self.L.clReleaseMemObject(b)

    def write(self, b, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes:
            rc = self.L.clEnqueueWriteBuffer(self.q, b, 1, offset, arr.nbytes, arr.ctypes.data_as(C.c_void_p), 0, None, None)
            assert rc == 0, f"clEnqueueWriteBuffer: {rc}"

    def read(self, b, arr, offset=0):
        assert arr.flags["C_CONTIGUOUS"]
        if arr.nbytes:
            rc = self.L.clEnqueueReadBuffer(self.q, b, 1, offset, arr.nbytes, arr.ctypes.data_as(C.c_void_p), 0, None, None)
            assert rc == 0, f"clEnqueueReadBuffer: {rc}"

    def launch(self, k, dims, args, timed=False):
        """args: buffer handles (c_void_p / int), python ints -> cl_uint, None -> a null image/buffer handle."""
        for i, a in enumerate(args):
            if isinstance(a, _U32):
                v = C.c_uint32(a.v)
                rc = self.L.clSetKernelArg(k, i, 4, C.byref(v))
            elif a is None:
                # an image2d_t the variant never samples: no image object can be created on this device (CL_DEVICE_IMAGE_SUPPORT = 0),
                # so offer the runtime a null handle; if it refuses, the caller switches to the hip backend
                h = C.c_void_p(0)
                rc = self.L.clSetKernelArg(k, i, C.sizeof(C.c_void_p), C.byref(h))
                if rc != 0:
                    raise ImageArgUnsupported(f"clSetKernelArg({i}, null image) = {rc}: the kernel takes an image2d_t and this device has no image support")
            else:
                h = C.c_void_p(a)
                rc = self.L.clSetKernelArg(k, i, C.sizeof(C.c_void_p), C.byref(h))
            assert rc == 0, f"clSetKernelArg({i}): {rc}"
        g = (C.c_size_t * len(dims))(*dims)
        ev = C.c_void_p()
        rc = self.L.clEnqueueNDRangeKernel(self.q, k, len(dims), None, g, None, 0, None, C.byref(ev) if timed else None)   # local size: NullRange, as the reference
        assert rc == 0, f"clEnqueueNDRangeKernel: {rc}"
        if not timed:
            return None
        self.L.clWaitForEvents(1, C.byref(ev))
        t0, t1 = C.c_uint64(), C.c_uint64()
        self.L.clGetEventProfilingInfo(ev, 0x1282, 8, C.byref(t0), None)       # CL_PROFILING_COMMAND_START
        self.L.clGetEventProfilingInfo(ev, 0x1283, 8, C.byref(t1), None)       # CL_PROFILING_COMMAND_END
        self.L.clReleaseEvent(ev)
        return (t1.value - t0.value) * 1e-6

    def finish(self):
        assert self.L.clFinish(self.q) == 0


class ImageArgUnsupported(RefGpuUnavailable):
    pass


class _U32:
    def __init__(self, v):
        self.v = int(v)


# ------------------------------------------------------------------------------------------------ HIP module backend
class _HipModule:
    name = "hip"

    def __init__(self):
        try:
            L = C.CDLL("libamdhip64.so")
        except OSError as e:
            raise RefGpuUnavailable(f"libamdhip64.so not loadable: {e}")
        self.L = L
        vp = C.c_void_p
        L.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
        L.hipFree.argtypes = [vp]
        L.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
        L.hipMemset.argtypes = [vp, C.c_int, C.c_size_t]
        L.hipModuleLoadData.argtypes = [C.POINTER(vp), vp]
        L.hipModuleGetFunction.argtypes = [C.POINTER(vp), vp, C.c_char_p]
        L.hipModuleLaunchKernel.argtypes = [vp] + [C.c_uint] * 6 + [C.c_uint, vp, C.POINTER(vp), C.POINTER(vp)]
        L.hipDeviceSynchronize.argtypes = []
        L.hipEventCreate.argtypes = [C.POINTER(vp)]
        L.hipEventRecord.argtypes = [vp, vp]
        L.hipEventSynchronize.argtypes = [vp]
        L.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), vp, vp]
        L.hipGetErrorString.restype = C.c_char_p
        L.hipGetErrorString.argtypes = [C.c_int]
        n = C.c_int()
        if L.hipGetDeviceCount(C.byref(n)) != 0 or n.value == 0:
            raise RefGpuUnavailable("no HIP device")
        self._mods = {}
        self._keep = []
        self._ev = None

    def _chk(self, rc, what):
        if rc != 0:
            raise RefGpuUnavailable(f"{what}: hip error {rc} ({self.L.hipGetErrorString(rc).decode()})")

    def device_name(self):
        return "hip device 0"

    def load(self, path):
        if path in self._mods:
            return self._mods[path]
        data = open(path, "rb").read()
        buf = C.create_string_buffer(data, len(data))
        self._keep.append(buf)
        m = C.c_void_p()
        self._chk(self.L.hipModuleLoadData(C.byref(m), buf), f"hipModuleLoadData({os.path.basename(path)})")
        self._mods[path] = m
        return m

    def kernel(self, mod, name):
        f = C.c_void_p()
        self._chk(self.L.hipModuleGetFunction(C.byref(f), mod, name.encode()), f"hipModuleGetFunction({name})")
        return f

    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.L.hipMalloc(C.byref(p), max(16, int(nbytes))), "hipMalloc")
        self.L.hipMemset(p, 0, max(16, int(nbytes)))
        return p.value

    def free(self, p):
        if p:
            self.L.hipFree(C.c_void_p(p))

    def write(self, p, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes:
            self._chk(self.L.hipMemcpy(C.c_void_p(p + offset), arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1), "hipMemcpy H2D")

    def read(self, p, arr, offset=0):
        assert arr.flags["C_CONTIGUOUS"]
        if arr.nbytes:
            self._chk(self.L.hipMemcpy(arr.ctypes.data_as(C.c_void_p), C.c_void_p(p + offset), arr.nbytes, 2), "hipMemcpy D2H")

    def launch(self, f, dims, args, timed=False):
        # the reference enqueues with a NullRange local size; here 256 x 1 x 1 work-groups (64 if the range is not a multiple of 256).
        # Every kernel guards its global id (queue length / maxId / numTasks / width*height), so a padded last group is harmless.
        n = int(dims[0])
        ny = int(dims[1]) if len(dims) > 1 else 1
        bx = 256 if n % 256 == 0 else 64
        gx = (n + bx - 1) // bx
        vals = []
        for a in args:
            if isinstance(a, _U32):
                vals.append(C.c_uint32(a.v))
            elif a is None:
                vals.append(C.c_void_p(0))
            else:
                vals.append(C.c_void_p(a))
        ptrs = (C.c_void_p * len(vals))(*[C.cast(C.pointer(v), C.c_void_p) for v in vals])
        if timed:
            if self._ev is None:
                a, b = C.c_void_p(), C.c_void_p()
                self.L.hipEventCreate(C.byref(a)); self.L.hipEventCreate(C.byref(b))
                self._ev = (a, b)
            self.L.hipEventRecord(self._ev[0], None)
        self._chk(self.L.hipModuleLaunchKernel(f, gx, ny, 1, bx, 1, 1, 0, None, ptrs, None), "hipModuleLaunchKernel")
        if timed:
            self.L.hipEventRecord(self._ev[1], None)
            self.L.hipEventSynchronize(self._ev[1])
            ms = C.c_float()
            self.L.hipEventElapsedTime(C.byref(ms), self._ev[0], self._ev[1])
            return float(ms.value)
        return None

    def finish(self):
        self._chk(self.L.hipDeviceSynchronize(), "hipDeviceSynchronize")


_backends = {}


def backend(name):
    if name not in _backends:
        _backends[name] = {"opencl": _OpenCL, "hip": _HipModule}[name]()
    return _backends[name]


# ------------------------------------------------------------------------------------------------ the context
class RefGpuContext:
    """The reference's wavefront kernels (gfx950 code objects) behind the interface of OracleContext / HipContext."""
    name = "reference-gfx950"

    def __init__(self, num_tasks, backend_name="opencl", flavour="ieee"):
        if not available(flavour):
            raise RefGpuUnavailable(f"{co_dir(flavour)} not built (make -C oracle/ref gfx950, needs /root/reference)")
        self.B = backend(backend_name)
        self.flavour = flavour
        self.num_tasks = N = int(num_tasks)
        self.params = None
        self.npix = 0
        B = self.B
        self.tasks = B.alloc(NUM_COLS * N * 4)
        self.queues = [B.alloc(N * 4) for _ in range(8)]
        self.counters = B.alloc(32)
        self.curr_pixel = B.alloc(4)
        self.params_buf = B.alloc(256)
        self.host_pixel_idx = 0
        self.fb = [None] * 6                      # pixels, preview, denAlbedoGL, denNormalGL, denAlbedo, denNormal (read_pixels order)
        self.scene = {}
        for k in ("tris", "nodes", "indices", "materials", "texdesc", "texdata"):
            self.scene[k] = B.alloc(256)
        # dummy 1-entry env tables (src/clcontext.cpp:513-518 creates dummies too)
        self.prob, self.alias, self.pdf = B.alloc(16), B.alloc(16), B.alloc(16)
        B.write(self.prob, np.ones(1, np.float32)); B.write(self.pdf, np.ones(1, np.float32)); B.write(self.alias, np.zeros(1, np.int32))
        B.write(self.counters, np.zeros(8, np.uint32)); B.write(self.curr_pixel, np.zeros(1, np.uint32))
        B.write(self.tasks, np.zeros(NUM_COLS * N, np.float32))
        self.env_img = None                       # {w, h, 0, 0, float4 texels[]} for the image stand-in builds of `logic` (upload_envmap)
        self._k = {}
        self.last_ms = {}
        self.timed = False

    # ---- kernels
    def _kernel(self, file, entry=None):
        key = (file, entry or file)
        if key not in self._k:
            prog = self.B.load(os.path.join(co_dir(self.flavour), file + ".co"))
            self._k[key] = self.B.kernel(prog, entry or file)
        return self._k[key]

    def _run(self, name, k, dims, args):
        ms = self.B.launch(k, dims, args, timed=self.timed)
        if ms is not None:
            self.last_ms[name] = ms

    def close(self):
        B = self.B
        for b in [self.tasks, self.counters, self.curr_pixel, self.params_buf, self.prob, self.alias, self.pdf, self.env_img] + self.queues + list(self.scene.values()) + self.fb:
            B.free(b)
        self.tasks = None

    # ---- uploads
    def upload_scene(self, d):
        B = self.B
        for k, arr in (("tris", d.tris), ("nodes", d.nodes), ("indices", d.indices), ("materials", d.materials), ("texdesc", d.texdesc), ("texdata", d.texdata)):
            B.free(self.scene[k])
            self.scene[k] = B.alloc(max(256, arr.nbytes))
            B.write(self.scene[k], arr)

    def upload_envmap(self, e):
        """gfx950 has no image support (CL_DEVICE_IMAGE_SUPPORT = 0), so no runtime can create the image2d_t `logic` expects.  In the
        logic_v<id>_imgstandin.co builds the kernel's 8-byte image slot is tagged a plain global pointer (oracle/ref/Makefile GERULE, last step), and they (the reference's
        wf_logic.cl unmodified, AMD's built-in library for everything EXCEPT read_imagef / get_image_dim, which come from the builder-written
        oracle/ref/gfx950_image_standin.cl) read {int w, h, 0, 0; float4 texels[w * h]} behind it.  A stand-in for the image filter only; labelled so."""
        if not available_env(self.flavour):
            raise RefGpuUnavailable(f"{co_dir(self.flavour)}/logic_v*_imgstandin.co not built (make -C oracle/ref gfx950)")
        B = self.B
        w, h = int(e.w), int(e.h)
        rgb = np.ascontiguousarray(e.rgb, np.float32).reshape(w * h, 3)
        img = np.zeros(4 + 4 * w * h, np.float32)
        img[:4].view(np.int32)[:2] = (w, h)
        tex = img[4:].reshape(w * h, 4)
        tex[:, :3] = rgb; tex[:, 3] = 1.0
        for b in (self.env_img, self.prob, self.alias, self.pdf):
            B.free(b)
        self.env_img = B.alloc(img.nbytes); B.write(self.env_img, img)
        self.prob, self.alias, self.pdf = B.alloc(4 * w * h), B.alloc(4 * w * h), B.alloc(4 * w * h)
        B.write(self.prob, np.ascontiguousarray(e.prob, np.float32)); B.write(self.alias, np.ascontiguousarray(e.alias, np.int32)); B.write(self.pdf, np.ascontiguousarray(e.pdf, np.float32))

    def set_params(self, p):
        self.params = p.copy()
        raw = np.frombuffer(np.ascontiguousarray(self.params).reshape(1).tobytes(), np.uint8)
        assert raw.size == 240
        self.B.write(self.params_buf, raw)
        npix = int(p["width"]) * int(p["height"])
        if npix != self.npix:
            for b in self.fb:
                self.B.free(b)
            self.fb = [self.B.alloc(npix * 16) for _ in range(6)]
            z = np.zeros(npix * 4, np.float32)
            for b in self.fb:
                self.B.write(b, z)
            self.npix = npix

    def set_partition(self, rank, nranks):
        assert (rank, nranks) == (0, 1), "the reference has no partition"

    def set_option(self, name, value):
        raise AssertionError(name)

    # ---- the wavefront kernels: NDRanges of src/clcontext.cpp:765-850, argument order of src/wf_*.cl
    def wf_reset(self):
        n = max(self.num_tasks, self.npix)
        pixels, denAlbedo, denNormal = self.fb[0], self.fb[4], self.fb[5]
        self._run("reset", self._kernel("reset"), [n], [self.tasks, pixels, denAlbedo, denNormal, self.counters, self.queues[Q_RAYGEN], self.params_buf, _U32(self.num_tasks)])

    def wf_raygen(self):
        self._run("genRays", self._kernel("genRays"), [self.num_tasks],
                  [self.tasks, self.params_buf, self.counters, self.queues[Q_RAYGEN], self.queues[Q_EXTENSION], self.curr_pixel, _U32(self.num_tasks)])

    def _trace(self, entry, q):
        s = self.scene
        self._run(entry, self._kernel(entry), [self.num_tasks],
                  [self.tasks, self.counters, self.queues[q], s["tris"], s["nodes"], s["indices"], self.params_buf, _U32(self.num_tasks)])

    def wf_extend(self): self._trace("traceExtension", Q_EXTENSION)
    def wf_shadow(self): self._trace("traceShadow", Q_SHADOW)

    def logic_variant(self):
        p = self.params       # build flags: src/kernel_impl.hpp:49-67
        return (1 if p["useAreaLight"] else 0) | (2 if p["useEnvMap"] else 0) | (4 if p["sampleExpl"] else 0) | (8 if p["sampleImpl"] else 0) | (16 if not p["wfSeparateQueues"] else 0)

    def wf_logic(self, first=False):
        v = self.logic_variant()
        if (v & 2) and self.env_img is None:
            raise ImageArgUnsupported("logic with USE_ENV_MAP samples an image: on gfx950 only through the image stand-in build (hip loader + upload_envmap)")
        n = ((self.num_tasks - 1) // 32 + 1) * 32
        s, q = self.scene, self.queues
        pixels, denAlbedo, denNormal = self.fb[0], self.fb[4], self.fb[5]
        args = [self.tasks, pixels, denNormal, denAlbedo, self.counters, q[Q_EXTENSION], q[Q_SHADOW], q[Q_RAYGEN], q[Q_DIFFUSE], q[Q_GLOSSY], q[Q_GGX_REFL],
                q[Q_GGX_REFR], q[Q_DELTA], s["tris"], s["nodes"], s["indices"], (self.env_img if (v & 2) else None), self.prob, self.alias, self.pdf, s["materials"], s["texdata"], s["texdesc"],
                self.params_buf, _U32(self.num_tasks), _U32(1 if first else 0)]
        self._run("logic", self._kernel(f"logic_v{v}_imgstandin" if (v & 2) else f"logic_v{v}", "logic"), [n], args)

    def _mat(self, entry, q):
        s = self.scene
        self._run(entry, self._kernel(entry), [self.num_tasks],
                  [self.tasks, self.counters, self.queues[q], self.queues[Q_EXTENSION], s["materials"], s["texdata"], s["texdesc"], self.params_buf, _U32(self.num_tasks)])

    def wf_materials(self):
        if self.params["wfSeparateQueues"]:          # src/clcontext.cpp:796-813
            self._mat("wavefrontDiffuse", Q_DIFFUSE); self._mat("wavefrontGlossy", Q_GLOSSY); self._mat("wavefrontGGXReflection", Q_GGX_REFL)
            self._mat("wavefrontGGXRefraction", Q_GGX_REFR); self._mat("wavefrontDelta", Q_DELTA)
        else:
            self._mat("wavefrontAllMaterials", Q_DIFFUSE)

    def postprocess(self):
        pixels, preview, denAlbedoGL, denNormalGL, denAlbedo, denNormal = self.fb
        self._run("process", self._kernel("process"), [self.npix], [pixels, denAlbedo, denNormal, preview, denAlbedoGL, denNormalGL, self.params_buf, _U32(self.num_tasks)])

    # ---- host-side bookkeeping (src/clcontext.cpp:877-900)
    def clear_queues(self): self.B.write(self.counters, np.zeros(8, np.uint32))
    def finish(self): self.B.finish()

    def get_counters(self):
        out = np.zeros(8, np.uint32)
        self.B.read(self.counters, out)
        return out

    def set_counters(self, c): self.B.write(self.counters, np.ascontiguousarray(c, np.uint32))

    def pixel_index_update(self, npix, nnew):
        self.host_pixel_idx = (self.host_pixel_idx + int(nnew)) % int(npix)
        self.B.write(self.curr_pixel, np.array([self.host_pixel_idx], np.uint32))

    def pixel_index_reset(self):
        self.host_pixel_idx = 0
        self.B.write(self.curr_pixel, np.zeros(1, np.uint32))

    def read_pixels(self, which=0):
        out = np.zeros((self.npix, 4), np.float32)
        self.B.read(self.fb[which], out)
        return out

    def write_pixels(self, which, arr):
        self.B.write(self.fb[which], np.ascontiguousarray(arr, np.float32))

    def state_export(self):
        out = np.zeros((NUM_COLS, self.num_tasks), np.float32)
        self.B.read(self.tasks, out)
        return out

    def state_import(self, st):
        st = np.ascontiguousarray(st, np.float32)
        assert st.shape == (NUM_COLS, self.num_tasks)
        self.B.write(self.tasks, st)

    def queue_read(self, q):
        out = np.zeros(self.num_tasks, np.uint32)
        self.B.read(self.queues[q], out)
        return out

    def queue_write(self, q, arr):
        self.B.write(self.queues[q], np.ascontiguousarray(arr, np.uint32))
