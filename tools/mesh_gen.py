"""A synthetic triangle-mesh scene in the reference's scene-file grammar (scene.h:340-427: obj_beg / obj_vtx / obj_tri / obj_end,
instanced with translate / scale / rotate): SURVEY.md 8(f) N2's "triangle-heavy scene".  Used by tests/ (through tests/_oracle.py),
the developer scripts here and bench.py's `configs.mesh`.  Data only - nothing of the product or the oracle is imported."""
import numpy as np


def mesh_scene(path, nu=16, nv=32, camera="camera 6 2.5 7 0 0.8 0 0 1 0 35 0.05 9"):
    """A UV sphere of nu x nv quads as ONE obj, instanced three times (translated, scaled, rotated), over a
    field of small spheres on the usual ground: 3 x 960 triangles for the defaults."""
    rng = np.random.default_rng(9)
    lines = [camera, "material a lambertian 0.6 0.5 0.4", "material m metal 0.8 0.8 0.9 0.05", "material g dielectric 1.5",
             "material r lambertian 0.8 0.2 0.2", "sphere 0 -1000 0 1000 a"]
    for i in range(-5, 5):
        for j in range(-5, 5):
            lines.append("sphere %r 0.2 %r 0.2 %s" % (i + 0.9 * float(rng.uniform()), j + 0.9 * float(rng.uniform()), "amg"[(i + j) % 3]))
    verts = [(np.sin(np.pi * i / nu) * np.cos(2 * np.pi * j / nv), np.cos(np.pi * i / nu), np.sin(np.pi * i / nu) * np.sin(2 * np.pi * j / nv)) for i in range(nu + 1) for j in range(nv)]
    tris = []
    for i in range(nu):
        for j in range(nv):
            a, b, c, d = i * nv + j, i * nv + (j + 1) % nv, (i + 1) * nv + j, (i + 1) * nv + (j + 1) % nv
            if i > 0:
                tris.append((a, c, b))
            if i < nu - 1:
                tris.append((b, c, d))
    lines.append("obj_beg %d %d" % (len(verts), len(tris)))
    lines += ["obj_vtx %r %r %r" % tuple(float(x) for x in v) for v in verts]
    lines += ["obj_tri %d %d %d" % t for t in tris]
    lines += ["obj_end", "obj 0 r t 0 1.0 0", "obj 0 m s 0.6 0.6 0.6 t 2.2 0.6 1.5", "obj 0 g s 0.7 0.5 0.7 r 40 0 0 1 t -2.0 0.7 1.0"]
    open(path, "w").write("\n".join(lines) + "\n")
    return str(path), 3 * len(tris)
