"""In-tree builds of the PRODUCT (no JIT caches): libfluctus_hip.so (hipcc, gfx950) and libfluctus_host.so (g++).
The checker (oracle/) has its own build entry, oracle/build.py; nothing here knows about it."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fluctus_amd")

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
             "-munsafe-fp-atomics", "-fgpu-rdc" if False else "-fno-gpu-rdc"]
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _run(cmd, cwd=ROOT):
    print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stdout.write(r.stdout)
        raise RuntimeError(f"build failed: {' '.join(cmd)}")
    return r.stdout


def _headers():
    return glob.glob(os.path.join(ROOT, "include", "*.h"))


def build_host(force=False):
    src = sorted(glob.glob(os.path.join(PKG, "host", "*.cpp")))
    dep = src + glob.glob(os.path.join(PKG, "host", "*.hpp")) + _headers() + [os.path.join(PKG, "csrc", "flx_wide.h")]   # host_capi.cpp includes the wide-tree builder
    out = os.path.join(PKG, "libfluctus_host.so")
    if force or _stale(out, dep):
        _run(["g++"] + CXX_FLAGS + ["-fopenmp"] + src + ["-o", out, "-ldl", "-lz"])
    return out


def build_hip(force=False):
    src = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
    dep = src + glob.glob(os.path.join(PKG, "csrc", "*.h")) + _headers()
    out = os.path.join(PKG, "libfluctus_hip.so")
    if force or _stale(out, dep):
        hipcc = "/opt/rocm/bin/hipcc"
        _run([hipcc] + [f for f in HIP_FLAGS if f] + src + ["-o", out])
    return out


def source_hash():
    """sha256 (first 16 hex digits) over the sources libfluctus_hip.so is built from: csrc/*.hip, csrc/*.h, include/*.h (names + contents).
    A PMC capture (profiles/traffic_<workload>.json) carries the hash of the kernels it was taken on; bench.py quotes it only while it matches."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.h")) + _headers())
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_all(force=False):
    """Build whatever is stale.  Serialised with a file lock: with `torch.distributed.run` every rank calls this at start-up."""
    import fcntl
    if os.environ.get("FLX_NO_BUILD") == "1":
        return []
    with open(os.path.join(PKG, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return [build_hip(force), build_host(force)]
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
