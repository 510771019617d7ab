"""A/B of device variants in one process: per-stage HIP-event times for the sponza-class 1080p frame."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magr_ray_tracer_amd import scenes  # noqa: E402
from magr_ray_tracer_amd.renderer import Device  # noqa: E402

W, H = 1920, 1080
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variants = [int(x) for x in sys.argv[2:]] or [1, 2, 0]
s, view = scenes.sponza_class(1.0)
sa = s.arrays()
cam = scenes.camera_for(view, W, H)
ref = None
for rep in range(2):
    for v in variants:
        for accel in (int(os.environ.get("AB_ACCEL", "0")),):
            d = Device(W, H, accel=accel, profile=True, extend_variant=v)
            d.upload(sa)
            cam["focalLength"] = d.focus(W // 2, H // 2, cam)
            d.seed_default()
            d.render(cam, 2)
            d.synchronize()
            d.reset(); d.seed_default(); d.reset_counters(); d.reset_stage_times()
            t = time.perf_counter()
            d.render(cam, frames)
            d.synchronize()
            dt = time.perf_counter() - t
            st = d.stage_times()
            c = d.counters()
            a = d.read_accum()
            if ref is None:
                ref = a
            same = np.array_equal(a.view(np.uint32), ref.view(np.uint32))
            B = c["extend_rays"] * 48 + c["extend_inst_visits"] * 68 + c["extend_node_visits"] * 96 + c["extend_prim_tests"] * 52
            print(f"variant {v} accel {accel}: {dt / frames * 1e3:.3f} ms/frame  " +
                  "  ".join(f"{k[:-3]} {st[k] / frames:.3f}" for k in st if k.endswith("_ms")) +
                  f"  extend {B / (st['extend_ms'] * 1e-3) / 1e9:.0f} GB/s  bit-equal-to-first {same} crc {int(a.view(np.uint32).astype(np.uint64).sum())}", flush=True)
            d.close()
