"""Write a SceneData (triangle soup + materials) as a pbrt-v3 scene: one binary PLY mesh per material plus a .pbrt file whose
materials are the inverse of the reference's PBRT->Fluctus mapping (src/scene.cpp:729-806).  Harness utility: it lets the
procedural stand-in scenes travel through the same ingest path a real PBRT scene takes (host/pbrt.cpp)."""
import os
import numpy as np
from .wire import BXDF


def _write_ply(path, tris):
    """binary_little_endian PLY, unshared vertices (x y z nx ny nz u v), triangle faces."""
    n = tris.size
    v = np.zeros((n, 3, 8), np.float32)
    for k, name in enumerate(("v0", "v1", "v2")):
        for j, f in enumerate(("p", "n")):
            for c, ax in enumerate("xyz"):
                v[:, k, 3 * j + c] = tris[name][f][ax]
        v[:, k, 6] = tris[name]["t"]["x"]
        v[:, k, 7] = tris[name]["t"]["y"]
    faces = np.zeros(n, np.dtype([("n", "u1"), ("i", "<i4", 3)]))
    faces["n"] = 3
    faces["i"] = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "property float nx\nproperty float ny\nproperty float nz\nproperty float u\nproperty float v\n"
           "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (3 * n, n))
    with open(path, "wb") as f:
        f.write(hdr.encode()); f.write(v.tobytes()); f.write(faces.tobytes())


def _rgb(v):
    return "[ %.9g %.9g %.9g ]" % (float(v["x"]), float(v["y"]), float(v["z"]))


def _material(m):
    t = int(m["type"])
    rough = 1.0 - float(m["Ns"]) / 5000.0                       # inverse of (1 - r) * 5000 with remaproughness true
    if t == BXDF.DIFFUSE:
        return 'Material "matte" "rgb Kd" %s' % _rgb(m["Kd"])
    if t == BXDF.GLOSSY:                                        # uber carries its own index, plastic / substrate force 1.5
        return 'Material "uber" "rgb Kd" %s "rgb Ks" %s "float roughness" %.9g "float index" %.9g' % (_rgb(m["Kd"]), _rgb(m["Ks"]), rough, float(m["Ni"]))
    if t == BXDF.GGX_ROUGH_REFLECTION:
        ni = float(m["Ni"])
        return 'Material "metal" "rgb eta" [ %.9g %.9g %.9g ] "rgb k" %s "float roughness" %.9g' % (ni, ni, ni, _rgb(m["Ks"]), rough)
    if t == BXDF.IDEAL_REFLECTION:
        return 'Material "mirror" "rgb Kr" %s' % _rgb(m["Ks"])
    if t == BXDF.IDEAL_DIELECTRIC:
        return 'Material "glass" "rgb Kt" %s "float index" %.9g' % (_rgb(m["Ks"]), float(m["Ni"]))
    return None                                                 # GGX dielectric / emissive have no PBRT counterpart in the reference's mapping


def export(d, folder, name="scene", camera=((0, 1, 5), (0, 1, 0), (0, 1, 0))):
    """Returns the .pbrt path and the list of material ids that could not be expressed (their triangles are written as matte)."""
    os.makedirs(folder, exist_ok=True)
    eye, look, up = camera
    lines = ["LookAt %g %g %g  %g %g %g  %g %g %g" % (*eye, *look, *up), 'Camera "perspective" "float fov" [ 60 ]', "WorldBegin"]
    skipped = []
    order = []                                                  # first use, like the reference numbers them
    for mid in d.tris["matId"]:
        if mid not in order:
            order.append(int(mid))
        if len(order) == d.materials.size:
            break
    for mid in order:
        sel = d.tris[d.tris["matId"] == mid]
        ply = "%s_mat%d.ply" % (name, mid)
        _write_ply(os.path.join(folder, ply), sel)
        mat = _material(d.materials[mid]) if mid > 0 else None
        lines.append("AttributeBegin")
        if mat is None and mid > 0:
            skipped.append(mid)
            mat = 'Material "matte" "rgb Kd" %s' % _rgb(d.materials[mid]["Kd"])
        if mat:
            lines.append("  " + mat)
        lines.append('  Shape "plymesh" "string filename" "%s"' % ply)
        lines.append("AttributeEnd")
    lines.append("WorldEnd")
    path = os.path.join(folder, name + ".pbrt")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path, skipped
