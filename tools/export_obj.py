"""Writes the procedural atrium as a Wavefront OBJ (+ MTL with one textured material + its PNG) so that the OBJ path can be exercised at
scale: `python tools/export_obj.py /tmp/atrium 1.0 && python bench.py --model /tmp/atrium/atrium.obj --view=-15,3.2,0.6,-0.97,-0.1,-0.05,75`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from magr_ray_tracer_amd import scenes  # noqa: E402
from magr_ray_tracer_amd.scene import save_png  # noqa: E402

out, detail = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
os.makedirs(out, exist_ok=True)
s, view = scenes.sponza_class(detail)
sa = s.arrays(bvh4=False)
p = sa.prims[:-2]                                  # without the light quad (model_scene adds the reference's own)
tri = np.stack([p["v0"][:, :3], p["v1"][:, :3], p["v2"][:, :3]], axis=1)
yy, xx = np.mgrid[0:64, 0:64]
tex = np.zeros((64, 64, 4), np.float32)
tex[..., 0] = 0.55 + 0.35 * ((xx // 8 + yy // 8) % 2); tex[..., 1] = 0.5 + 0.2 * ((xx // 8) % 2); tex[..., 2] = 0.42
save_png(os.path.join(out, "paving.png"), tex)
with open(os.path.join(out, "atrium.mtl"), "w") as f:
    f.write("newmtl paving\nKd 1 1 1\nmap_Kd paving.png\nnewmtl plain\nKd 0.8 0.8 0.8\n")
with open(os.path.join(out, "atrium.obj"), "w") as f:
    f.write("# procedural atrium (magr_ray_tracer_amd.scenes.sponza_class)\nmtllib atrium.mtl\n")
    np.savetxt(f, tri.reshape(-1, 3), fmt="v %.6f %.6f %.6f")
    uv = np.stack([tri[..., 0] * 0.25, tri[..., 2] * 0.25], axis=-1).reshape(-1, 2)
    np.savetxt(f, uv - np.floor(uv), fmt="vt %.6f %.6f")
    n = len(tri)
    idx = np.arange(1, 3 * n + 1).reshape(n, 3)
    floor = np.abs(tri[..., 1]).max(axis=1) < 0.05          # the paving gets the texture
    f.write("usemtl paving\n")
    np.savetxt(f, np.concatenate([idx[floor], idx[floor]], axis=1)[:, [0, 3, 1, 4, 2, 5]], fmt="f %d/%d %d/%d %d/%d")
    f.write("usemtl plain\n")
    np.savetxt(f, idx[~floor], fmt="f %d %d %d")
print(n, "triangles,", int(floor.sum()), "textured ->", os.path.join(out, "atrium.obj"), os.path.getsize(os.path.join(out, "atrium.obj")) >> 20, "MB")
