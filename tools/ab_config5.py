"""A/B on the config-5 scene (2 BLAS under a TLAS, SBVH alpha 0) at 1920x1080: variant 2 (nested loops, one ray per lane) vs 0 (persistent)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magr_ray_tracer_amd import scenes
from magr_ray_tracer_amd.renderer import Device
W, H, frames = 1920, 1080, 8
alpha = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
s, view = scenes.config5_scene(alpha)
print("alpha", alpha, s.stats())
sa = s.arrays()
cam = scenes.camera_for(view, W, H)
ref = None
for rep in range(2):
    for v in (2, 0):
        d = Device(W, H, profile=True, extend_variant=v)
        d.upload(sa)
        cam["focalLength"] = d.focus(W // 2, H // 2, cam)
        d.seed_default(); d.render(cam, 2); d.synchronize()
        d.reset(); d.seed_default(); d.reset_counters(); d.reset_stage_times()
        t = time.perf_counter(); d.render(cam, frames); d.synchronize(); dt = time.perf_counter() - t
        st = d.stage_times(); a = d.read_accum()
        if ref is None: ref = a
        print(f"config5 variant {v}: {dt / frames * 1e3:.3f} ms/frame  " + "  ".join(f"{k[:-3]} {st[k] / frames:.3f}" for k in st if k.endswith("_ms")) +
              f"  bit-equal-to-first {np.array_equal(a.view(np.uint32), ref.view(np.uint32))}", flush=True)
        d.close()
