"""A REAL reference asset end to end: assets/egyptcat/egyptcat.obj -- scene #1 of the reference's own benchmark
(src/tracer.cpp:384-389) -- as committed fixtures (build container only: reads /root/reference).

  python scripts/make_egyptcat_fixture.py

  tests/golden/egyptcat_scene.npz   the scene exactly as host/scene.cpp's OBJ + MTL loader and host/texture.cpp's PNG decoder produce it
                                    from egyptcat.obj / egyptcat.mtl / EgyptCat.png (pinned against the reference's vendored tinyobj in
                                    tests/test_host.py): wire triangles (160 B each), materials (80 B), texture descriptor + RGBA8 blob.
                                    Data only -- no reference source text.
  tests/golden/steps_egyptcat.npz   outputs of the REFERENCE's own wf_*.cl kernels (oracle/_ref) on that scene with the reference's start-up
                                    parameters (src/tracer.cpp:38-52, 760-797: default camera and area light, no env map, 10 bounces, single
                                    material queue): state / queues / counters after every kernel of two iterations, in the format of the
                                    other steps_*.npz fixtures; the scene arrays are referenced (`scene_file`), the SBVH (host/bvh.cpp) is stored.
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from fluctus_amd import host, wire  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SRC = "/root/reference/assets/egyptcat/egyptcat.obj"


def main():
    d = host.load_scene(SRC)
    np.savez_compressed(os.path.join(OUT, "egyptcat_scene.npz"), tris=d.tris.view(np.uint8).reshape(-1), materials=d.materials.view(np.uint8).reshape(-1),
                        texdesc=d.texdesc.view(np.uint8).reshape(-1), texdata=d.texdata,
                        source=np.array("assets/egyptcat/egyptcat.obj + egyptcat.mtl + EgyptCat.png of harskish/fluctus, through fluctus_amd/host/scene.cpp"))
    print("egyptcat_scene.npz", d.tris.size, "triangles,", d.materials.size, "materials,", d.texdata.size, "texture bytes")
    import make_golden
    host.build_bvh(d, "sbvh")
    make_golden.steps("egyptcat", scene=d, scene_file="egyptcat_scene.npz", w=64, h=64, n=2048, params="reference")
    for f in ("egyptcat_scene.npz", "steps_egyptcat.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
