"""Builds of the CHECKER (test infrastructure): oracle/liboracle.so (g++) and -- only where /root/reference exists, i.e. in the build
container -- oracle/_ref: the reference's own kernels for x86-64 (libfluctus_ref.so, libfluctus_refbvh.so) and for gfx950
(oracle/_ref/gfx950/{ieee,fast}/*.co, run by ROCm's OpenCL runtime on the GPU box; oracle/ref_gpu.py).
Called by __graft_entry__.build() / smoke(), tests/conftest.py and bench.py's cpu_baseline leg -- never by fluctus_amd (the product's
build entry, fluctus_amd/build.py, builds the product and nothing else)."""
import fcntl
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd):
    print("[oracle build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stdout.write(r.stdout)
        raise RuntimeError(f"build failed: {' '.join(cmd)}")


def build_oracle(force=False):
    src = [os.path.join(HERE, "wf_oracle.cpp")]
    out = os.path.join(HERE, "liboracle.so")
    if force or _stale(out, src + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp"] + src + ["-o", out])
    return out


def build_ref(force=False):
    """The reference's own kernels (x86-64 and gfx950) -- only where the reference checkout exists (this container)."""
    if not os.path.isdir("/root/reference/src"):
        return None
    out = os.path.join(HERE, "_ref", "libfluctus_ref.so")
    marker = os.path.join(HERE, "_ref", "gfx950", "ieee", "traceExtension.co")
    dep = glob.glob(os.path.join(HERE, "ref", "*"))
    if force or _stale(out, dep):
        _run(["make", "-C", os.path.join(HERE, "ref"), "-j8"])
    if force or _stale(marker, dep):
        # the gfx950 leg depends on the image's device-libs layout: a failure here is reported, not fatal (the x86 pin and the CPU tests do not need it;
        # tests/test_gpu_ref_gfx950.py then fails loudly on the GPU box instead of silently passing)
        try:
            _run(["make", "-C", os.path.join(HERE, "ref"), "-j8", "gfx950"])
        except RuntimeError as e:
            print(f"[oracle build] WARNING: gfx950 build of the reference kernels failed ({e}); Pin 5 will be unavailable", flush=True)
    return out


def build_all(force=False):
    if os.environ.get("FLX_NO_BUILD") == "1":
        return []
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return [build_oracle(force), build_ref(force)]
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
