"""Minimal ctypes binding of the OpenCL 1.2 host API (libOpenCL.so.1 = the ICD loader; on the GPU box ROCm's runtime exposes
the MI355X) -- just enough to build a program from source and run 1-D kernels over numpy buffers.
TEST INFRASTRUCTURE for tests/test_gpu_ocl_builtins.py; nothing in the product uses OpenCL."""
import ctypes as C
import numpy as np

CL_DEVICE_TYPE_ALL = 0xFFFFFFFF
CL_DEVICE_NAME, CL_DEVICE_VERSION, CL_DEVICE_IMAGE_SUPPORT = 0x102B, 0x102F, 0x1016
CL_MEM_READ_WRITE, CL_MEM_COPY_HOST_PTR = 1, 32
CL_PROGRAM_BUILD_LOG = 0x1183


class OpenCLUnavailable(RuntimeError):
    pass


class Device:
    def __init__(self):
        try:
            L = C.CDLL("libOpenCL.so.1")
        except OSError as e:
            raise OpenCLUnavailable(f"libOpenCL.so.1 not loadable: {e}")
        self.L = L
        vp, u32 = C.c_void_p, C.c_uint32
        L.clCreateContext.restype = vp
        L.clCreateContext.argtypes = [vp, u32, C.POINTER(vp), vp, vp, C.POINTER(C.c_int)]
        L.clCreateCommandQueue.restype = vp
        L.clCreateCommandQueue.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_int)]
        L.clCreateProgramWithSource.restype = vp
        L.clCreateProgramWithSource.argtypes = [vp, u32, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.clBuildProgram.argtypes = [vp, u32, C.POINTER(vp), C.c_char_p, vp, vp]
        L.clGetProgramBuildInfo.argtypes = [vp, vp, u32, C.c_size_t, vp, C.POINTER(C.c_size_t)]
        L.clCreateKernel.restype = vp
        L.clCreateKernel.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
        L.clCreateBuffer.restype = vp
        L.clCreateBuffer.argtypes = [vp, C.c_uint64, C.c_size_t, vp, C.POINTER(C.c_int)]
        L.clSetKernelArg.argtypes = [vp, u32, C.c_size_t, vp]
        L.clEnqueueNDRangeKernel.argtypes = [vp, vp, u32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), u32, vp, vp]
        L.clEnqueueReadBuffer.argtypes = [vp, vp, u32, C.c_size_t, C.c_size_t, vp, u32, vp, vp]
        L.clFinish.argtypes = [vp]
        L.clReleaseMemObject.argtypes = [vp]
        L.clGetDeviceInfo.argtypes = [vp, u32, C.c_size_t, vp, C.POINTER(C.c_size_t)]
        L.clGetPlatformIDs.argtypes = [u32, C.POINTER(vp), C.POINTER(u32)]
        L.clGetDeviceIDs.argtypes = [vp, C.c_uint64, u32, C.POINTER(vp), C.POINTER(u32)]
        nplat = u32()
        if L.clGetPlatformIDs(0, None, C.byref(nplat)) != 0 or nplat.value == 0:
            raise OpenCLUnavailable("no OpenCL platform")
        plats = (vp * nplat.value)()
        L.clGetPlatformIDs(nplat.value, plats, None)
        self.dev = None
        for p in plats:
            nd = u32()
            if L.clGetDeviceIDs(p, CL_DEVICE_TYPE_ALL, 0, None, C.byref(nd)) != 0 or nd.value == 0:
                continue
            devs = (vp * nd.value)()
            L.clGetDeviceIDs(p, CL_DEVICE_TYPE_ALL, nd.value, devs, None)
            self.dev = vp(devs[0])
            break
        if self.dev is None:
            raise OpenCLUnavailable("OpenCL platform present but it exposes 0 devices")
        err = C.c_int()
        self.ctx = L.clCreateContext(None, 1, C.byref(self.dev), None, None, C.byref(err))
        if err.value != 0:
            raise OpenCLUnavailable(f"clCreateContext failed: {err.value}")
        self.q = L.clCreateCommandQueue(self.ctx, self.dev, 0, C.byref(err))
        if err.value != 0:
            raise OpenCLUnavailable(f"clCreateCommandQueue failed: {err.value}")

    def info_str(self, what):
        buf = C.create_string_buffer(512)
        self.L.clGetDeviceInfo(self.dev, what, 512, buf, None)
        return buf.value.decode(errors="replace")

    def image_support(self):
        v = C.c_uint32()
        self.L.clGetDeviceInfo(self.dev, CL_DEVICE_IMAGE_SUPPORT, 4, C.byref(v), None)
        return bool(v.value)

    def build(self, source, options=""):
        err = C.c_int()
        src = C.c_char_p(source.encode())
        prog = self.L.clCreateProgramWithSource(self.ctx, 1, C.byref(src), None, C.byref(err))
        assert err.value == 0, err.value
        rc = self.L.clBuildProgram(prog, 1, C.byref(self.dev), options.encode(), None, None)
        if rc != 0:
            n = C.c_size_t()
            self.L.clGetProgramBuildInfo(prog, self.dev, CL_PROGRAM_BUILD_LOG, 0, None, C.byref(n))
            log = C.create_string_buffer(n.value + 1)
            self.L.clGetProgramBuildInfo(prog, self.dev, CL_PROGRAM_BUILD_LOG, n.value, log, None)
            raise RuntimeError(f"clBuildProgram failed ({rc}):\n{log.value.decode(errors='replace')}")
        return Program(self, prog)


class Program:
    def __init__(self, dev, prog):
        self.d, self.prog = dev, prog

    def run(self, kernel, n, args):
        """args: numpy arrays (device buffers, copied in, ALL copied back in place after the launch) or python ints (cl_int)."""
        L, d = self.d.L, self.d
        err = C.c_int()
        k = L.clCreateKernel(self.prog, kernel.encode(), C.byref(err))
        assert err.value == 0, (kernel, err.value)
        bufs = []
        for i, a in enumerate(args):
            if isinstance(a, np.ndarray):
                assert a.flags["C_CONTIGUOUS"]
                b = L.clCreateBuffer(d.ctx, CL_MEM_READ_WRITE | CL_MEM_COPY_HOST_PTR, max(4, a.nbytes), a.ctypes.data_as(C.c_void_p), C.byref(err))
                assert err.value == 0, err.value
                bufs.append((b, a))
                h = C.c_void_p(b)
                assert L.clSetKernelArg(k, i, C.sizeof(C.c_void_p), C.byref(h)) == 0
            else:
                v = C.c_int(int(a))
                assert L.clSetKernelArg(k, i, 4, C.byref(v)) == 0
        g = C.c_size_t(n)
        rc = L.clEnqueueNDRangeKernel(d.q, k, 1, None, C.byref(g), None, 0, None, None)
        assert rc == 0, rc
        for b, a in bufs:
            assert L.clEnqueueReadBuffer(d.q, b, 1, 0, a.nbytes, a.ctypes.data_as(C.c_void_p), 0, None, None) == 0
        assert L.clFinish(d.q) == 0
        for b, _ in bufs:
            L.clReleaseMemObject(b)
