"""Throughput of the five BASELINE.json configurations on ONE MI355X (configs 4 and 5 are 8-GPU jobs there; this is one rank's share of
them at full frame size).  Parity of every configuration is covered by tests/ (config 1: test_config1_cube_256_cpu_plumbing, 2:
test_config2_bunny_class_720p_kajiya, 3: test_full_size_properties_1080p, 4: test_config4_bvh4_1080p_band_vs_oracle, 5:
test_config5_robo_orb_terrarium_tlas_sbvh / test_config5_4k_frame_runs_and_is_deterministic); this script only times them."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magr_ray_tracer_amd import scenes  # noqa: E402
from magr_ray_tracer_amd.renderer import Device  # noqa: E402

CONFIGS = [
    ("1 cube 256x256 (GPU path; the CPU plumbing run is a test)", scenes.cube_scene, 256, 256, 64, dict()),
    ("2 bunny-class 70k tris, 1280x720, 64 spp, Kajiya", lambda: scenes.bunny_class(187), 1280, 720, 64, dict(shading=0)),
    ("3 sponza-class SAH BVH2, 1920x1080, 256 spp, NEE", lambda: scenes.sponza_class(1.0), 1920, 1080, 256, dict()),
    ("4 sponza-class QBVH, 1920x1080, 128 of 1024 spp (one rank of 8)", lambda: scenes.sponza_class(1.0), 1920, 1080, 128, dict(accel=1)),
    ("5 robo-orb + terrarium_bot, 2 BLAS + TLAS, SBVH alpha 0, 3840x2160, 64 of 4096 spp", lambda: scenes.config5_scene(0.0), 3840, 2160, 64, dict()),
]
for name, fn, W, H, spp, kw in CONFIGS:
    t0 = time.perf_counter()
    s, view = fn()
    sa = s.arrays()
    tb = time.perf_counter() - t0
    cam = scenes.camera_for(view, W, H)
    d = Device(W, H, **kw)
    d.upload(sa)
    cam["focalLength"] = d.focus(W // 2, H // 2, cam)
    d.seed_default()
    d.render(cam, 2)
    d.synchronize()
    d.reset(); d.seed_default(); d.reset_counters()
    t = time.perf_counter()
    d.render(cam, spp)
    d.synchronize()
    dt = time.perf_counter() - t
    c = d.counters()
    print(f"config {name}: {len(sa.prims)} prims, host scene+BVH {tb:.2f} s, {spp} spp in {dt * 1e3:.1f} ms = {dt / spp * 1e3:.3f} ms/spp, "
          f"{W * H * spp / dt / 1e6:.1f} M samples/s, {(c['extend_rays'] + c['connect_rays']) / dt / 1e6:.0f} M traced rays/s", flush=True)
    d.close()
