"""Multi-BLAS scenes whose stack column needs 13..22 entries: spill (cap 12) + world ray in LDS, or the whole column in LDS without the backup?
usage (GPU box): python tools/middepth_tlas.py   - runs each policy in a child process (the library reads its switches at upload)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch  # noqa: F401  (the fast runtime stack, EXPERIMENTS.md (44))
    from magr_ray_tracer_amd import scenes
    from magr_ray_tracer_amd.renderer import Device
    alpha, n = float(sys.argv[2]), int(sys.argv[3])
    s, view = scenes.two_blas_scene(alpha, n)
    sa = s.arrays()
    W, H = 1920, 1080
    cam = scenes.camera_for(view, W, H)
    d = Device(W, H)
    d.upload(sa)
    k = d.kernel_info()
    d.seed_default()
    d.render(cam, 8); d.synchronize()
    t0 = time.perf_counter()
    d.render(cam, 64); d.synchronize()
    dt = time.perf_counter() - t0
    print(f"  alpha {alpha} n {n}: {len(sa.prims)} prims, persist {k['persist']} stack_entries {k['stack_entries']}: {W * H * 64 / dt / 1e6:.1f} M samples/s", flush=True)
    d.close()
    sys.exit(0)

for alpha, n in ((1.0, 48), (1.0, 160), (0.0, 48)):
    for env in ({"RT355_TLAS_BACKUP": "1"}, {"RT355_TLAS_BACKUP": "0"}, {"RT355_TLAS_BACKUP": "0", "RT355_NO_SPILL": "1"}, {"RT355_TLAS_BACKUP": "1", "RT355_NO_SPILL": "1"}):
        print(env, flush=True)
        subprocess.run([sys.executable, __file__, "child", str(alpha), str(n)], env=dict(os.environ, **env))
