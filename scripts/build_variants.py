"""Build A/B variants of libfluctus_hip.so with different -D tunables into gpurun-visible build/variants/.
usage: python scripts/build_variants.py name1:-DLDS_LEVELS=16,-DTRACE_MIN_WAVES=2 name2:...
Select at run time with FLX_HIP_LIB=<path>."""
import glob, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluctus_amd import build
out = os.path.join(ROOT, "variants")
os.makedirs(out, exist_ok=True)
src = sorted(glob.glob(os.path.join(ROOT, "fluctus_amd", "csrc", "*.hip")))


def one(spec):
    name, _, flags = spec.partition(":")
    flags = [f for f in flags.split(",") if f]
    dst = os.path.join(out, f"libfluctus_hip_{name}.so")
    cmd = ["/opt/rocm/bin/hipcc"] + build.HIP_FLAGS + flags + src + ["-o", dst]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return name, r.returncode, r.stdout[-2000:] if r.returncode else dst


with ThreadPoolExecutor(4) as ex:
    for name, rc, msg in ex.map(one, sys.argv[1:]):
        print(name, "OK" if rc == 0 else "FAILED", msg)
