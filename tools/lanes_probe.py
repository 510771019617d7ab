"""Do M independent contexts (own stream, own queues, own seed slice) of ONE process overlap on the GPU like M processes do?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magr_ray_tracer_amd import scenes  # noqa: E402
from magr_ray_tracer_amd.renderer import Device  # noqa: E402

W, H = 1920, 1080
s, view = scenes.sponza_class(1.0)
sa = s.arrays()
cam = scenes.camera_for(view, W, H)
frames = 256
for M in (1, 2, 3, 4, 1):
    devs = []
    for m in range(M):
        d = Device(W, H)
        d.upload(sa)
        d.seed_default()
        d.render(cam, 2)
        devs.append(d)
    for d in devs:
        d.synchronize()
    per = frames // M
    t = time.perf_counter()
    for f in range(per):
        for d in devs:
            d.render(cam, 1)
    for d in devs:
        d.synchronize()
    dt = time.perf_counter() - t
    print(f"lanes {M}: {per * M} frames in {dt * 1e3:.1f} ms = {dt / (per * M) * 1e3:.3f} ms/frame, {W * H * per * M / dt / 1e6:.1f} M samples/s", flush=True)
    for d in devs:
        d.close()
