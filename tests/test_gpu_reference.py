"""GPU suite (-m gpu): the REFERENCE's own OpenCL kernels (oracle/_ref, compiled from /root/reference by
oracle/build_ref.sh and shipped as machine code only) run on the MI355X next to the HIP path and the oracle.
extend() is compared bit for bit on identical rays; shade()/connect() under schedule S0 (one work-item)."""
import numpy as np
import pytest

import ref_gpu
from magr_ray_tracer_amd import _lib as W, scenes
from magr_ray_tracer_amd.renderer import Device
from oracle.oracle_py import Oracle, S0, seed_stream
from helpers import DEFAULT, assert_bits, max_rel

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_gpu.available(), reason="oracle/_ref not built (needs /root/reference at build time)")]
RW, RH = ref_gpu.REF_W, ref_gpu.REF_H


@pytest.mark.parametrize("accel", [0, 1])
def test_extend_hip_vs_reference_kernel_bit_exact(accel):
    v = dict(DEFAULT, accel=accel)
    rows = 32
    s, view = scenes.sponza_class(0.3)
    view = dict(view, forward=(-0.97, 0.55, -0.05))     # pitch so that the top rows of the frame see geometry
    sa = s.arrays()
    cam = scenes.camera_for(view, RW, RH)
    ref = ref_gpu.RefGPU(sa, **v)
    n = RW * rows
    seeds = seed_stream(0, n)
    gen, gseeds = ref.generate(cam, seeds)
    ext = ref.extend(gen)
    d = Device(RW, RH, y0=0, y1=rows, **v)
    d.upload(sa)
    d.set_rays(0, gen)
    d.stage_extend(0)
    got = d.get_rays(0)
    hit = ext["primIdx"] != -1
    assert hit.mean() > 0.5
    for f in ("t", "primIdx", "I", "N"):
        assert_bits(got[f], ext[f], "extend " + f)
    assert_bits(got["u"][hit], ext["u"][hit], "extend u")
    assert_bits(got["v"][hit], ext["v"][hit], "extend v")
    # and the HIP generate differs from the reference's only through normalize()'s hardware rsqrt
    d.set_seeds(seeds)
    d.stage_begin_frame()
    d.stage_generate(cam)
    mine = d.get_rays(0)
    assert_bits(mine["O"], gen["O"], "generate O")
    assert np.abs(mine["D"] - gen["D"]).max() < 1e-6      # unit-scale direction: a few ulp
    assert np.array_equal(d.get_seeds(), gseeds)
    d.close()
    ref.close()


def test_focus_matches_reference_kernel():
    s, view = scenes.mixed_scene()
    sa = s.arrays()
    cam = scenes.camera_for(view, RW, RH)
    ref = ref_gpu.RefGPU(sa, **DEFAULT)
    d = Device(RW, RH, **DEFAULT)
    d.upload(sa)
    for (x, y) in ((640, 360), (100, 600), (1200, 80), (640, 700), (5, 5)):
        a, b = ref.focus(x, y, cam), d.focus(x, y, cam)
        assert abs(a - b) <= 2e-6 * abs(a), (x, y, a, b)
    d.close()
    ref.close()


@pytest.mark.parametrize("vi", [
    dict(),                                                                         # NEE, cosine, RR, firefly filter (BASELINE config 3)
    dict(shading=0, sampling=0, russian_roulette=False, filter_fireflies=False),    # Kajiya, hemisphere (config 2 family)
    dict(sampling=0, russian_roulette=False),                                       # NEE, hemisphere, fixed path length
    dict(shading=0, sampling=1),                                                    # Kajiya, cosine, RR, firefly filter
    dict(filter_fireflies=False),                                                   # NEE without the firefly clamp
    dict(accel=1),                                                                  # NEE over the BVH4
])
def test_shade_connect_s0_oracle_vs_reference_kernels(vi):
    v = dict(DEFAULT, **vi)
    rows = 8
    s, view = scenes.mixed_scene()
    view = dict(view, forward=(0.32, 0.75, 0.92))
    sa = s.arrays()
    cam = scenes.camera_for(view, RW, RH)
    ref = ref_gpu.RefGPU(sa, **v)
    o = Oracle(sa, RW, RH, **v, schedule=S0)
    n = RW * rows
    gen, gseeds = ref.generate(cam, seed_stream(0, n))
    ext = ref.extend(gen)
    ref.clear_accum()
    rout, rsh, rseeds = ref.shade_s0(ext, gseeds)
    racc = ref.read_accum(rows)
    acc = np.zeros((RH * RW, 4), np.float32)
    seeds = gseeds.copy()
    out, sh = o.shade(ext.copy(), acc, seeds)
    assert len(out) == len(rout) and len(sh) == len(rsh) and seeds[0] == rseeds[0]
    for f in ("pixelIdx", "bounces", "inside", "lastSpecular"):
        assert np.array_equal(out[f], rout[f]), f
    assert max_rel(out["D"], rout["D"], 1e-2) < 1e-4 and max_rel(out["intensity"], rout["intensity"], 1e-3) < 1e-5
    assert max_rel(acc[:n], racc.reshape(-1, 4), 1e-4) < 1e-5
    if len(rsh):
        assert np.array_equal(sh["lightIdx"], rsh["lightIdx"]) and np.array_equal(sh["pixelIdx"], rsh["pixelIdx"])
        ref.clear_accum()
        ref.connect_s0(rsh)
        acc = np.zeros((RH * RW, 4), np.float32)
        o.connect(rsh.copy(), acc)
        assert max_rel(acc[:n], ref.read_accum(rows).reshape(-1, 4), 1e-4) < 1e-5
    ref.close()
