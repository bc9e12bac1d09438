"""tests/golden/night_env.npz: the reference's own environment map, assets/env_maps/night.hdr (512 x 256, the map SURVEY 8(d) names for the kitchen
and courtyard configurations), as DATA for the GPU box, where /root/reference does not exist.
    python scripts/make_envmap_fixture.py          (build container only)
The pixels are read by the REFERENCE's Radiance reader (src/rgbe/rgbe.cpp compiled in oracle/_ref; tests/test_host.py pins host/envmap.cpp's reader
to it bit for bit); the three sampling tables are host/envmap.cpp's (restatement of src/envmap.cpp:31-114) and are stored as a digest only -- the
fixture's loader (fluctus_amd.host.envmap_from_rgb) rebuilds them and tests/test_host.py checks the digest."""
import ctypes as C
import hashlib
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluctus_amd import host          # noqa: E402
from oracle import binding as ob      # noqa: E402

SRC = "/root/reference/assets/env_maps/night.hdr"


def main():
    L = ob.ref_lib()
    w, h = C.c_int(), C.c_int()
    assert L.ref_read_hdr(SRC.encode(), C.byref(w), C.byref(h), None) == 0
    rgb = np.zeros(w.value * h.value * 3, np.float32)
    assert L.ref_read_hdr(SRC.encode(), C.byref(w), C.byref(h), rgb.ctypes.data_as(C.c_void_p)) == 0
    e = host.load_envmap(SRC)
    assert (e.w, e.h) == (w.value, h.value) and np.array_equal(e.rgb, rgb)
    e2 = host.envmap_from_rgb(e.w, e.h, rgb)
    assert np.array_equal(e2.prob, e.prob) and np.array_equal(e2.alias, e.alias) and np.array_equal(e2.pdf, e.pdf)
    digest = hashlib.sha256(e.prob.tobytes() + e.alias.tobytes() + e.pdf.tobytes()).hexdigest()
    out = os.path.join(ROOT, "tests", "golden", "night_env.npz")
    np.savez_compressed(out, w=np.int32(e.w), h=np.int32(e.h), rgb=rgb, tables_sha256=np.array(digest), pdf0=np.float32(e.pdf[0]),
                        source=np.array("harskish/fluctus assets/env_maps/night.hdr (read by the reference's src/rgbe/rgbe.cpp)"))
    print(out, os.path.getsize(out), "bytes; tables sha256", digest, "pdf[0]", float(e.pdf[0]))


if __name__ == "__main__":
    main()
