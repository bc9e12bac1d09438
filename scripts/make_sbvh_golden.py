"""Golden digests of the SBVH builder's output from tests/sbvh_restatement.py (the second, independent restatement of the reference's
src/sbvh.cpp in Python over fp32 scalars) on meshes too large for the test suite to rebuild every run: teapot.ply (3 206 triangles, ~1.5 min
in Python) is rebuilt live by the test, the 38 k-triangle conference-proc mesh the SAH pin uses (~half an hour) is pinned through the
digest written here.

  python scripts/make_sbvh_golden.py          -> tests/golden/sbvh_restatement_digest.json

The fixture holds no reference source: SHA-256 of the node array (reference wire layout, 48 B per node) and of the index list the
restatement produces, plus its split / duplicate counts."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fluctus_amd import host, wire          # noqa: E402
from sbvh_restatement import SBVH           # noqa: E402


def restatement_arrays(d):
    P = np.stack([np.stack([d.tris[v]["p"][a] for a in "xyz"], -1) for v in ("v0", "v1", "v2")], 1).astype(np.float32)
    ref = SBVH(P)
    nodes = np.zeros(len(ref.nodes), wire.NODE)
    for i, (box, parent, link, nprims) in enumerate(ref.nodes):
        for k, a in enumerate("xyz"):
            nodes[i]["bmin"][a] = np.float32(box.mn[k]); nodes[i]["bmax"][a] = np.float32(box.mx[k])
        nodes[i]["parent"] = parent; nodes[i]["iStartOrRight"] = link; nodes[i]["nPrims"] = nprims
    return ref, nodes, np.array(ref.indices, np.uint32)


def digest(nodes, indices):
    cols = np.stack([nodes["bmin"]["x"], nodes["bmin"]["y"], nodes["bmin"]["z"], nodes["bmax"]["x"], nodes["bmax"]["y"], nodes["bmax"]["z"]], 1).astype(np.float32)
    meta = np.stack([nodes["parent"].astype(np.int64), nodes["iStartOrRight"].astype(np.int64), nodes["nPrims"].astype(np.int64)], 1)
    return {"boxes_sha256": hashlib.sha256(cols.tobytes()).hexdigest(), "links_sha256": hashlib.sha256(meta.tobytes()).hexdigest(),
            "indices_sha256": hashlib.sha256(indices.astype(np.uint32).tobytes()).hexdigest(), "nodes": int(nodes.size), "indices": int(indices.size)}


if __name__ == "__main__":
    out = {}
    for name, make in (("conference-38k", lambda: host.generate_scene("conference", 6000, 43)),):
        d = make()
        t0 = time.time()
        ref, nodes, idx = restatement_arrays(d)
        e = digest(nodes, idx)
        e.update({"triangles": int(d.tris.size), "splits": int(ref.splits), "spatial": int(ref.spatial), "duplicates": int(ref.duplicates), "depth": int(ref.depth),
                  "restatement_seconds": round(time.time() - t0, 1)})
        out[name] = e
        print(name, e, flush=True)
    path = os.path.join(ROOT, "tests", "golden", "sbvh_restatement_digest.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)
