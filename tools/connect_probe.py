"""GPU box: what do the connect launches of a bench frame consist of?  Shadow rays per bounce, fraction occluded, and the any-hit
work (node + triangle records) spent on occluded vs unoccluded rays (oracle counters on a sample band)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magr_ray_tracer_amd import scenes  # noqa: E402
from magr_ray_tracer_amd.renderer import Device  # noqa: E402

W, H = 1920, 1080
s, view = scenes.sponza_class(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
sa = s.arrays(bvh4=False)
cam = scenes.camera_for(view, W, H)
d = Device(W, H)
d.upload(sa)
cam["focalLength"] = d.focus(W // 2, H // 2, cam)
d.seed_default()
d.stage_begin_frame()
d.stage_generate(cam)
for b in range(7):
    d.stage_extend(b)
    d.stage_shade(b)
tot = occ = 0
for b in range(7):
    before = d.get_shadow(b, b)
    n = len(before)
    tot += n
    print(f"bounce {b}: {n} shadow rays", flush=True)
c0 = d.counters()
d.stage_connect(0, 6)
d.synchronize()
c1 = d.counters()
for b in range(7):
    after = d.get_shadow(b, b)
    z = int((np.abs(after["radiance"]).max(1) == 0).sum()) if len(after) else 0
    occ += z
    print(f"bounce {b}: {z} of {len(after)} contribute nothing after connect ({100.0 * z / max(len(after), 1):.1f} %)")
print("total", tot, "zero after connect", occ, f"({100.0 * occ / tot:.1f} %)")
print("connect records per ray:", (c1["connect_node_visits"] - c0["connect_node_visits"]) / tot, "nodes,",
      (c1["connect_prim_tests"] - c0["connect_prim_tests"]) / tot, "triangles")
