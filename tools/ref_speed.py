"""Times the REFERENCE's own OpenCL extend kernel (oracle/_ref code object) on the MI355X against the HIP extend on the same
primary rays (1280x720, sponza-class scene) — a like-for-like 'reference on this hardware' data point."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_gpu  # noqa: E402
from magr_ray_tracer_amd import scenes  # noqa: E402
from magr_ray_tracer_amd.renderer import Device  # noqa: E402
from oracle.oracle_py import seed_stream  # noqa: E402

RW, RH = ref_gpu.REF_W, ref_gpu.REF_H
s, view = scenes.sponza_class(1.0)
sa = s.arrays()
cam = scenes.camera_for(view, RW, RH)
ref = ref_gpu.RefGPU(sa)
n = RW * RH
gen, _ = ref.generate(cam, seed_stream(0, n))
B = None
for g in (2560, 256 * 256, 256 * 1024, 256 * 2048):
    ms = min(ref.extend_timed(gen, g) for _ in range(3))
    print(f"reference extend (OpenCL code object), {g:7d} persistent work-items: {ms:8.3f} ms  -> {n / ms / 1e3:8.1f} M rays/s", flush=True)
for v in (1, 2, 0):
    d = Device(RW, RH, profile=True, extend_variant=v)
    d.upload(sa)
    d.set_rays(0, gen)
    d.stage_extend(0)
    d.synchronize()
    d.reset_stage_times(); d.reset_counters()
    for _ in range(5):
        d.stage_extend(0)
    d.synchronize()
    st = d.stage_times()
    c = d.counters()
    ms = st["extend_ms"] / st["extend_launches"]
    bytes_ = (c["extend_rays"] * 48 + c["extend_inst_visits"] * 68 + c["extend_node_visits"] * 96 + c["extend_prim_tests"] * 52) / st["extend_launches"]
    print(f"HIP extend variant {v}: {ms:8.3f} ms -> {n / ms / 1e3:8.1f} M rays/s, {bytes_ / ms / 1e6:8.1f} GB/s algorithmic", flush=True)
    d.close()
