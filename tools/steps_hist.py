"""Per bounce: how long is the longest ray?  `steps` (the reference's heat-map count: children entered / pushed) of every ray of a frame.
usage (GPU box): python tools/steps_hist.py [config 2|3|5]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import bench  # noqa: E402
from magr_ray_tracer_amd import scenes  # noqa: E402
from magr_ray_tracer_amd.renderer import Device  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
conf = bench.CONFIGS[cfg]


class A:
    detail = 1.0


s, view = conf["scene"](A)
sa = s.arrays()
W, H = conf["W"], conf["H"]
cam = scenes.camera_for(view, W, H)
d = Device(W, H, shading=conf["shading"])
d.upload(sa)
cam["focalLength"] = d.focus(W // 2, H // 2, cam)
d.seed_default()
d.render(cam, 3)
d.enable_steps(True)
d.stage_begin_frame()
d.stage_generate(cam)
for b in range(7):
    d.stage_extend(b)
    n = len(d.get_rays(b))
    st = d.get_steps()
    if n is not None:
        st = st[:n]
    st = st[st >= 0]
    nz = st[st > 0]
    q = np.percentile(st, [50, 90, 99, 99.9, 99.99]) if len(st) else [0] * 5
    print(f"config {cfg} bounce {b}: rays {len(st)} steps mean {st.mean():.1f} p50 {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} p99.9 {q[3]:.0f} p99.99 {q[4]:.0f} max {st.max()}  rays above half the max: {(st > st.max() / 2).sum()}", flush=True)
    d.stage_shade(b)
d.close()
