"""CPU experiment driver: shadow rays of a bench-scene frame (from the oracle) -> tools/lab/anyhit_lab (visit counts per order)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from magr_ray_tracer_amd import scenes, _lib as Wl
from oracle.oracle_py import Oracle, seed_stream
W, H = 480, 270
s, view = scenes.sponza_class(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
sa = s.arrays(bvh4=False)
cam = scenes.camera_for(view, W, H)
o = Oracle(sa, W, H)
cam["focalLength"] = o.focus(W // 2, H // 2, cam)
n = W * H
seeds = seed_stream(0, n)
rays = o.generate(cam, 0, n, seeds)
acc = np.zeros((n, 4), np.float32)
sh_all = []
for b in range(7):
    o.extend(rays)
    rays, sh = o.shade(rays, acc, seeds)
    sh_all.append(sh)
sh = np.concatenate(sh_all)
rec = np.zeros(len(sh), dtype=np.dtype([("o", "<f4", 3), ("tmax", "<f4"), ("d", "<f4", 3), ("pix", "<i4")]))
rec["o"] = (sh["I"] + sh["L"] * np.float32(1e-4))[:, :3]
rec["tmax"] = sh["dist"] - np.float32(2e-4)
rec["d"] = sh["L"][:, :3]
rec["pix"] = sh["pixelIdx"]
path = "/tmp/anyhit_lab.bin"
with open(path, "wb") as f:
    np.array([len(sa.bvh2), len(sa.prims), len(sa.primIdx), len(rec)], np.int32).tofile(f)
    sa.bvh2.tofile(f); sa.prims.tofile(f); sa.primIdx.tofile(f); rec.tofile(f)
exe = "/tmp/anyhit_lab"
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tools/lab/anyhit_lab.c"), "-o", exe, "-lm"])
print(len(rec), "shadow rays")
subprocess.check_call([exe, path])
