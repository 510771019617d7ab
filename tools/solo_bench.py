"""One context, no torch in the process unless asked for: which HIP runtime stack is the fast one for a plain host?
usage: python tools/solo_bench.py [config 3|5] [--torch-first | --torch-after]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--torch-first" in sys.argv:
    import torch  # noqa: F401
from magr_ray_tracer_amd import scenes  # noqa: E402
from magr_ray_tracer_amd.renderer import Device  # noqa: E402
if "--torch-after" in sys.argv:
    import torch  # noqa: F401,E402
    torch.zeros(4, device="cuda")          # initialise torch's runtime as well

cfg = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
if cfg == 5:
    s, view = scenes.config5_scene(0.0); W, H = 3840, 2160
else:
    s, view = scenes.sponza_class(1.0); W, H = 1920, 1080
sa = s.arrays()
cam = scenes.camera_for(view, W, H)
d = Device(W, H)
d.upload(sa)
cam["focalLength"] = d.focus(W // 2, H // 2, cam)
d.seed_default()
d.render(cam, 4); d.synchronize()
t0 = time.perf_counter()
n = 96
d.render(cam, n); d.synchronize()
dt = time.perf_counter() - t0
libs = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln or "libhsa-runtime64" in ln})
print(f"config {cfg} one context {' '.join(a for a in sys.argv[2:])}: {W * H * n / dt / 1e6:.1f} M samples/s; runtimes mapped: {libs}")
