"""GPU parity suite (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs.  Everything is required to be BIT-EXACT (accumulator, ray queues, RNG state, work counters) - since
round 2 also the scenes that reach transcendentals (sphere lights, sphere textures, glass, fisheye): exp / sin / cos /
acos / atan2 are evaluated by the same Cephes-style sequences of IEEE operations on both sides (rt355_kernels.h rt_expf ...).
Only post-processing keeps a tolerance (pow and the hardware sqrt of the reference's length(), see test_gpu_reference.py)."""
import numpy as np
import pytest

from magr_ray_tracer_amd import _lib as W, scenes
from magr_ray_tracer_amd.renderer import Device, Renderer, RtError
from oracle.oracle_py import Oracle, seed_stream
from helpers import DEFAULT, assert_bits, bits_equal, build, max_rel

pytestmark = pytest.mark.gpu

TRI_SCENES = {
    "cube": scenes.cube_scene,
    "bunny32": lambda: scenes.bunny_class(32),
    "sponza.2": lambda: scenes.sponza_class(0.2),
}


def _pair(scene_fn, Wd, Hd, variant, y0=0, y1=None):
    s, sa, cam = build(scene_fn, Wd, Hd)
    o = Oracle(sa, Wd, Hd, **{k: v for k, v in variant.items() if k != "extend_variant"})
    d = Device(Wd, Hd, y0=y0, y1=y1, **variant)
    d.upload(sa)
    return sa, cam, o, d


def _ctr_equal(dev, e, c):
    """extend: every work counter equals the oracle's (same visit order as the reference).  connect is an any-hit traversal with
    its own visit order inside a BLAS (rt355_kernels.h slab_any): whether a ray is occluded - hence the accumulator - does not depend
    on it, the node / triangle counts do; rays, TLAS and instance visits are still the reference's."""
    for k in ("rays", "tlas_visits", "inst_visits", "node_visits", "prim_tests"):
        assert dev["extend_" + k] == e[k], ("extend_" + k, dev["extend_" + k], e[k])
    for k in ("rays", "tlas_visits", "inst_visits"):
        assert dev["connect_" + k] == c[k], ("connect_" + k, dev["connect_" + k], c[k])
    assert (dev["connect_node_visits"] > 0) == (c["node_visits"] > 0)


@pytest.mark.parametrize("scene", list(TRI_SCENES))
def test_stage_by_stage_bit_exact(scene):
    Wd, Hd = 96, 54
    sa, cam, o, d = _pair(TRI_SCENES[scene], Wd, Hd, DEFAULT)
    f = o.focus(Wd // 2, Hd // 2, cam)
    assert d.focus(Wd // 2, Hd // 2, cam) == f
    cam["focalLength"] = f
    n = Wd * Hd
    seeds = seed_stream(0, n)
    d.set_seeds(seeds.copy())
    d.enable_steps()
    acc = np.zeros((Hd * Wd, 4), np.float32)
    d.reset()
    d.stage_begin_frame()
    d.stage_generate(cam)
    rays = o.generate(cam, 0, n, seeds)
    g = d.get_rays(0)
    for fld in ("O", "D", "intensity", "pixelIdx", "bounces", "inside", "lastSpecular"):
        assert_bits(g[fld], rays[fld], "generate " + fld)
    assert np.array_equal(d.get_seeds(), seeds)
    shadows = []
    for b in range(W.MAX_BOUNCES):
        steps, _ = o.extend(rays, want_steps=True)
        d.stage_extend(b)
        g = d.get_rays(b)
        hit = rays["primIdx"] != -1
        for fld in ("t", "primIdx", "I", "N"):
            assert_bits(g[fld], rays[fld], f"extend{b} {fld}")
        assert_bits(g["u"][hit], rays["u"][hit], f"extend{b} u")
        assert_bits(g["v"][hit], rays["v"][hit], f"extend{b} v")
        assert np.array_equal(d.get_steps()[:len(rays)], steps), f"extend{b} steps"   # the reference's own `steps` value
        nxt, sh = o.shade(rays, acc, seeds)
        d.stage_shade(b)
        g = d.get_rays(b + 1)
        assert len(g) == len(nxt), f"shade{b}: queue length {len(g)} vs {len(nxt)}"
        for fld in ("O", "D", "intensity", "pixelIdx", "bounces", "inside", "lastSpecular"):
            assert_bits(g[fld], nxt[fld], f"shade{b} {fld}")
        assert np.array_equal(d.get_seeds(), seeds), f"shade{b} RNG state"
        rec = d.get_shadow(b, b)
        assert len(rec) == len(sh)
        if len(sh):
            eps = np.float32(1e-4)
            assert_bits(rec["o"], (sh["I"] + sh["L"] * eps)[:, :3], f"shadow{b} origin")
            assert_bits(rec["l"], sh["L"][:, :3], f"shadow{b} dir")
            assert_bits(rec["tmax"], sh["dist"] - np.float32(2) * eps, f"shadow{b} tmax")
            assert np.array_equal(rec["pixelIdx"], sh["pixelIdx"])
        shadows.append(sh)
        rays = nxt
    o.connect(np.concatenate(shadows), acc)
    d.stage_connect(0, W.MAX_BOUNCES - 1)
    assert_bits(d.read_accum().reshape(-1, 4), acc, "accumulator")
    d.close()


VARIANTS = [
    dict(),
    dict(extend_variant=1),      # traverse the reference arrays as uploaded (no derived layout)
    dict(extend_variant=2),      # derived layout, one ray per lane (no persistent wavefronts)
    dict(shading=0),
    dict(shading=0, sampling=0, russian_roulette=False, filter_fireflies=False),
    dict(sampling=0),
    dict(russian_roulette=False),
    dict(filter_fireflies=False),
    dict(accel=1),
    dict(accel=1, shading=0),
]


@pytest.mark.parametrize("vi", range(len(VARIANTS)))
def test_frames_bit_exact_all_variants(vi):
    v = dict(DEFAULT, **VARIANTS[vi])
    Wd, Hd, frames = 128, 72, 3
    sa, cam, o, d = _pair(lambda: scenes.sponza_class(0.2), Wd, Hd, v)
    cam["focalLength"] = o.focus(Wd // 2, Hd // 2, cam)
    acc, seeds, e, c = o.render(cam, frames)
    d.seed_default()
    d.render(cam, frames)
    assert_bits(d.read_accum(), acc, "accumulator")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    assert acc[..., :3].sum() > 0
    d.close()


@pytest.mark.parametrize("alpha,accel", [(0.0, 0), (0.0, 1), (1e-5, 0)])
def test_sbvh_and_tlas_two_blas(alpha, accel):
    v = dict(DEFAULT, accel=accel)
    Wd, Hd = 128, 72

    def fn():
        s, view = scenes.two_blas_scene(alpha=alpha, n=20)
        return s, view
    # the glass sphere reaches exp() (Beer's law): bit-exact like everything else (rt_expf / orc_expf)
    sa, cam, o, d = _pair(fn, Wd, Hd, v)
    assert len(sa.blas) == 2
    acc, seeds, e, c = o.render(cam, 2)
    d.seed_default()
    d.render(cam, 2)
    assert_bits(d.read_accum(), acc, f"two BLAS, alpha={alpha}, accel={accel}")
    assert np.array_equal(d.get_seeds(), seeds)
    dc = d.counters()
    assert dc["extend_tlas_visits"] > 0
    _ctr_equal(dc, e, c)
    d.close()


def test_tlas_triangles_only_bit_exact():
    def fn():
        from magr_ray_tracer_amd.scenes import Scene, _std_materials, param_surface, box_tris
        s = Scene()
        _std_materials(s)
        s.AddTriangles(box_tris((-2, 0, -1), (-0.5, 1.5, 0.5)), "red")
        s.AddQuad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), "grey")
        s.AddQuad((-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1), "white-light")
        s.BuildBLAS(0, 0.0)
        st = s.num_prims
        s.AddTriangles(box_tris((0.6, 0, -0.8), (2.0, 2.2, 0.8)), "mirror")
        s.AddTriangles(box_tris((0.2, 2.6, -0.4), (1.0, 3.0, 0.4)), "green")
        s.BuildBLAS(st, 0.0)
        st = s.num_prims
        s.AddTriangles(box_tris((-0.4, 0, 1.4), (0.4, 0.8, 2.2)), "sand")
        s.BuildBLAS(st, 1.0)
        return s, dict(origin=(0.5, 2.4, 6.0), forward=(0.05, 0.2, 0.97), fov=65.0, aperture=0.03)
    for accel in (0, 1):
        v = dict(DEFAULT, accel=accel)
        sa, cam, o, d = _pair(fn, 128, 72, v)
        assert len(sa.blas) == 3 and len(sa.tlas) == 6
        acc, seeds, e, c = o.render(cam, 3)
        d.seed_default()
        d.render(cam, 3)
        assert_bits(d.read_accum(), acc, f"accumulator accel={accel}")
        _ctr_equal(d.counters(), e, c)
        d.close()


def test_all_primitive_and_material_kinds_bit_exact():
    """Spheres (diffuse/mirror/glass/emissive), textured triangles, two light kinds: reaches sin/cos/exp/acos/atan2."""
    Wd, Hd, frames = 160, 90, 4
    sa, cam, o, d = _pair(scenes.mixed_scene, Wd, Hd, DEFAULT)
    cam["focalLength"] = o.focus(Wd // 2, Hd // 2, cam)
    acc, seeds, e, c = o.render(cam, frames)
    d.seed_default()
    d.render(cam, frames)
    assert_bits(d.read_accum(), acc, "mixed scene accumulator")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    d.close()


def test_row_band_matches_oracle_band():
    v = DEFAULT
    Wd, Hd = 128, 72
    sa, cam, o, d = _pair(lambda: scenes.sponza_class(0.2), Wd, Hd, v, y0=24, y1=48)
    acc, seeds, e, c = o.render(cam, 2, y0=24, y1=48)
    d.seed_default()
    d.render(cam, 2)
    got = d.read_accum()
    assert_bits(got, acc, "band accumulator")
    assert not got[:24].any() and not got[48:].any()
    d.close()


def test_frames_accumulate_and_reset():
    Wd, Hd = 96, 54
    sa, cam, o, d = _pair(scenes.cube_scene, Wd, Hd, DEFAULT)
    d.seed_default()
    d.render(cam, 1)
    a1 = d.read_accum()
    d.render(cam, 2)
    a3 = d.read_accum()
    ref, *_ = o.render(cam, 3)
    assert_bits(a3, ref, "1+2 frames == 3 frames")
    assert not bits_equal(a1, a3)
    d.reset()
    assert not d.read_accum().any()
    d.close()


def test_render_bvh_heat_map():
    Wd, Hd = 96, 54
    sa, cam, o, d = _pair(lambda: scenes.bunny_class(24), Wd, Hd, DEFAULT)
    seeds = seed_stream(0, Wd * Hd)
    rays = o.generate(cam, 0, Wd * Hd, seeds)
    steps, _ = o.extend(rays, want_steps=True)
    d.seed_default()
    d.render(cam, 1, renderBVH=1)
    got = d.read_accum().reshape(-1, 4)
    exp = (steps.astype(np.uint32).astype(np.float32) / np.float32(255.0))
    assert_bits(got[:, 0], exp, "steps/255 heat map")
    d.close()


def test_edge_cases_empty_and_tiny():
    # 1x1 frame, 64x1 band (exactly one wave), ragged sizes that are no multiple of the wave or block width
    for (Wd, Hd) in ((1, 1), (64, 1), (65, 3), (257, 2)):
        sa, cam, o, d = _pair(scenes.cube_scene, Wd, Hd, DEFAULT)
        acc, seeds, e, c = o.render(cam, 2)
        d.seed_default()
        d.render(cam, 2)
        assert_bits(d.read_accum(), acc, f"{Wd}x{Hd}")
        d.close()
    # a camera that sees nothing: every ray misses, queues run empty after bounce 0
    s, view = scenes.cube_scene()
    sa = s.arrays()
    cam = scenes.make_camera(64, 36, (0, 50, 0), (0, -1, 0.001), fov=40.0)
    o = Oracle(sa, 64, 36, **DEFAULT)
    d = Device(64, 36, **DEFAULT)
    d.upload(sa)
    acc, *_ = o.render(cam, 1)
    d.seed_default()
    d.render(cam, 1)
    assert_bits(d.read_accum(), acc, "all-miss frame")
    assert d.counters()["extend_rays"] == 64 * 36
    d.close()


def test_api_error_paths():
    d = Device(64, 36)
    with pytest.raises(RtError, match="no scene"):
        d.render(scenes.make_camera(64, 36, (0, 0, 0), (0, 0, 1)), 1)
    with pytest.raises(RtError, match="expected"):
        d.set_seeds(np.zeros(5, np.uint32))
    s, view = scenes.cube_scene()
    sa = s.arrays()
    bad = sa.primIdx.copy()
    bad[0] = 10 ** 6
    import copy
    sb = copy.copy(sa)
    sb.primIdx = bad
    with pytest.raises(RtError, match="primIdx"):
        d.upload(sb)
    with pytest.raises(RtError):
        Device(64, 36, y0=30, y1=20)
    d.close()


def test_renderer_mirror_init_tick():
    """The C++ Renderer mirror (Init / Tick / FocusCamera / ComputeEnergy) drives the same device path."""
    Wd, Hd = 96, 54
    s, view = scenes.cube_scene()
    r = Renderer(s, Wd, Hd)
    r.SetCamera(view["origin"], view["forward"], fov=view["fov"], aperture=view["aperture"])
    r.Init()
    r.Tick(3)
    img, energy = r.read()
    cam = r.camera()
    sa = s.arrays()
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    assert cam["focalLength"] == o.focus(Wd // 2, Hd // 2, cam)
    ref, *_ = o.render(cam, 3)
    assert_bits(img, ref, "Renderer::Tick x3")
    assert energy > 0
    r.close()


def test_full_size_properties_1080p():
    """BASELINE config 3 size (1920x1080): properties that do not need the CPU oracle at full size."""
    Wd, Hd = 1920, 1080
    s, view = scenes.sponza_class(0.5)
    sa = s.arrays()
    cam = scenes.camera_for(view, Wd, Hd)
    d = Device(Wd, Hd, **DEFAULT)
    d.upload(sa)
    cam["focalLength"] = d.focus(Wd // 2, Hd // 2, cam)
    d.seed_default()
    d.render(cam, 2)
    a = d.read_accum()
    s2 = d.get_seeds()
    c = d.counters()
    # determinism: a second context reproduces the image and RNG state bit for bit
    d2 = Device(Wd, Hd, **DEFAULT)
    d2.upload(sa)
    d2.seed_default()
    d2.render(cam, 1)
    d2.render(cam, 1)
    assert bits_equal(d2.read_accum(), a) and np.array_equal(d2.get_seeds(), s2)
    assert c["extend_rays"] >= 2 * Wd * Hd and c["primary_rays"] == 2 * Wd * Hd
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all() and a[..., :3].mean() > 0.01
    # an oracle spot check on a 16-row band of the same frame
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    ref, *_ = o.render(cam, 1, y0=520, y1=536)
    d3 = Device(Wd, Hd, y0=520, y1=536, **DEFAULT)
    d3.upload(sa)
    d3.seed_default()
    d3.render(cam, 1)
    assert_bits(d3.read_accum(), ref, "1080p band vs oracle")
    for x in (d, d2, d3):
        x.close()


def test_postproc_chain_matches_oracle():
    from oracle.oracle_py import postproc
    Wd, Hd = 160, 90
    sa, cam, o, d = _pair(scenes.cube_scene, Wd, Hd, DEFAULT)
    d.seed_default()
    d.render(cam, 3)
    acc = d.read_accum()
    # prep + chromatic: only + - * / fma -> bit-exact, float image and bytes.  vignetting goes through length(), which the reference's
    # kernel (and k_postproc, to match it bit for bit: test_postproc_chain_vs_reference_kernels) evaluates with the hardware v_sqrt_f32
    for vig, chroma in ((0.0, 0.0), (0.7, 0.0), (0.0, 0.15), (0.5, 0.05)):
        f, b = d.postproc(3, vignette=vig, gamma=1.0, chromatic=chroma)
        ef, eb = postproc(acc, 3, vig, 1.0, chroma)
        if vig == 0.0:
            assert_bits(f, ef, f"postproc float vig={vig} chroma={chroma}")
            assert np.array_equal(b, eb)
        else:
            assert max_rel(f, ef, 1e-6) < 3e-6 and np.abs(b.astype(int) - eb.astype(int)).max() <= 1
    # default gamma 0.9 goes through pow(): tolerance on floats, bytes may differ by one code
    f, b = d.postproc(3, vignette=0.3, gamma=0.9, chromatic=0.05)
    ef, eb = postproc(acc, 3, 0.3, 0.9, 0.05)
    assert np.abs(f - ef).max() < 1e-5 and np.abs(b.astype(int) - eb.astype(int)).max() <= 1
    d.close()


def test_renderer_save_frame_png(tmp_path):
    s, view = scenes.cube_scene()
    r = Renderer(s, 64, 36)
    r.SetCamera(view["origin"], view["forward"], fov=view["fov"], aperture=view["aperture"])
    r.Init()
    r.Tick(2)
    p = tmp_path / "frame.png"
    r.SaveFrame(p)
    raw = p.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and len(raw) > 64 * 36 * 3
    r.close()


def test_config2_bunny_class_720p_kajiya():
    """BASELINE config 2: ~70k-triangle closed mesh, binary SAH BVH, 1280x720, SHADING_SIMPLE ("Kajiya") — full frame vs oracle."""
    Wd, Hd, frames = 1280, 720, 2
    v = dict(DEFAULT, shading=0)
    sa, cam, o, d = _pair(lambda: scenes.bunny_class(187), Wd, Hd, v)
    assert 69000 < len(sa.prims) < 71000
    cam["focalLength"] = o.focus(Wd // 2, Hd // 2, cam)
    assert d.focus(Wd // 2, Hd // 2, cam) == cam["focalLength"]
    acc, seeds, e, c = o.render(cam, frames, threads=1)
    d.seed_default()
    d.render(cam, frames)
    assert_bits(d.read_accum(), acc, "config 2 accumulator")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    assert c["rays"] == 0                      # Kajiya: no shadow rays
    d.close()


def test_config5_robo_orb_terrarium_tlas_sbvh():
    """BASELINE config 5 geometry (robo-orb + terrarium_bot, 2 BLAS + TLAS, SBVH alpha = 0, glass dome) at reduced size.
    The glass dome reaches exp() (Beer's law): bit-exact, counters equal."""
    Wd, Hd, frames = 256, 144, 2
    sa, cam, o, d = _pair(lambda: scenes.config5_scene(0.0), Wd, Hd, DEFAULT)
    assert len(sa.prims) == 35600 + 40012 + 4 and len(sa.primIdx) > len(sa.prims)
    cam["focalLength"] = o.focus(Wd // 2, Hd // 2, cam)
    acc, seeds, e, c = o.render(cam, frames)
    d.seed_default()
    d.render(cam, frames)
    assert_bits(d.read_accum(), acc, "config 5 scene accumulator")
    assert np.array_equal(d.get_seeds(), seeds)
    dc = d.counters()
    _ctr_equal(dc, e, c)
    assert dc["extend_tlas_visits"] > 0
    d.close()


def test_checkpoint_resume_is_bit_exact(tmp_path):
    """{accum, seeds, frames} is the whole cross-frame state: 2 frames + checkpoint + 2 frames in a new context == 4 frames."""
    Wd, Hd = 96, 54
    sa, cam, o, d = _pair(lambda: scenes.sponza_class(0.2), Wd, Hd, DEFAULT)
    d.seed_default()
    d.render(cam, 2)
    ck = tmp_path / "ck.npz"
    d.save_checkpoint(ck, frames=3)
    d.close()
    d2 = Device(Wd, Hd, **DEFAULT)
    d2.upload(sa)
    assert d2.load_checkpoint(ck) == 3
    d2.render(cam, 2)
    ref, seeds, *_ = o.render(cam, 4)
    assert_bits(d2.read_accum(), ref, "resumed accumulator")
    assert np.array_equal(d2.get_seeds(), seeds)
    d2.close()


def test_camera_controller_resets_accumulation():
    s, view = scenes.cube_scene()
    r = Renderer(s, 64, 36)
    r.SetCamera(view["origin"], view["forward"], fov=view["fov"], aperture=view["aperture"])
    r.Init()
    r.Tick(3)
    assert r.frames() == 4                       # frames starts at 1 and counts up (renderer.cpp:45,53)
    a3, _ = r.read()
    c0 = r.camera()
    r.Move(3)                                    # CamDir::Right: origin += right * speed
    r.Tick(1)
    c1 = r.camera()
    assert np.allclose(c1["origin"][:3] - c0["origin"][:3], c0["right"][:3], atol=1e-6)
    assert r.frames() == 2                       # camera.moved -> reset kernel, frames = 1, then ++
    a1, _ = r.read()
    assert a1[..., :3].sum() < 0.6 * a3[..., :3].sum()
    r.Zoom(-10.0)
    r.Tick(1)
    assert abs(float(r.camera()["fov"]) - (view["fov"] - 10.0)) < 1e-5 and r.frames() == 2
    r.close()


def test_config4_bvh4_1080p_band_vs_oracle():
    """BASELINE config 4 shape: sponza-class scene through the BVH4 collapse at 1920x1080, one of 8 row bands (rank 3 of 8)."""
    from magr_ray_tracer_amd import dist as rdist
    Wd, Hd = 1920, 1080
    v = dict(DEFAULT, accel=1)
    p = rdist.plan("bands", Wd, Hd, 3, 8)
    s, view = scenes.sponza_class(0.5)
    sa = s.arrays()
    cam = scenes.camera_for(view, Wd, Hd)
    o = Oracle(sa, Wd, Hd, **v)
    d = Device(Wd, Hd, y0=p["y0"], y1=p["y1"], **v)
    d.upload(sa)
    cam["focalLength"] = d.focus(Wd // 2, Hd // 2, cam)
    ref, seeds, e, c = o.render(cam, 1, y0=p["y0"], y1=p["y1"], threads=1, seeds=seed_stream(p["seed_first"], p["seed_count"]))
    d.set_seeds(seed_stream(p["seed_first"], p["seed_count"]))
    d.render(cam, 1)
    assert_bits(d.read_accum(), ref, "config 4 band accumulator")
    _ctr_equal(d.counters(), e, c)
    d.close()


def test_config5_4k_frame_runs_and_is_deterministic():
    """BASELINE config 5 size: 3840x2160 (8.3 M pixels, ~4 GB of queues) on the two-BLAS scene; full-size properties."""
    Wd, Hd = 3840, 2160
    s, view = scenes.config5_scene(0.0)
    sa = s.arrays()
    cam = scenes.camera_for(view, Wd, Hd)
    imgs = []
    for _ in range(2):
        d = Device(Wd, Hd, **DEFAULT)
        d.upload(sa)
        cam["focalLength"] = d.focus(Wd // 2, Hd // 2, cam)
        d.seed_default()
        d.render(cam, 1)
        imgs.append(d.read_accum())
        c = d.counters()
        d.close()
    assert bits_equal(imgs[0], imgs[1])
    assert c["primary_rays"] == Wd * Hd and c["extend_rays"] > Wd * Hd and c["extend_tlas_visits"] == c["extend_rays"]
    assert np.isfinite(imgs[0]).all() and imgs[0][..., :3].mean() > 0.01
    # spot check against the oracle on an 8-row band through the middle of the frame
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    ref, *_ = o.render(cam, 1, y0=1076, y1=1084)
    d = Device(Wd, Hd, y0=1076, y1=1084, **DEFAULT)
    d.upload(sa)
    d.seed_default()
    d.render(cam, 1)
    assert_bits(d.read_accum(), ref, "config 5 at 4K: 8-row band vs oracle")
    d.close()


def test_fewer_bounces_no_aa_and_fisheye():
    Wd, Hd = 128, 72
    # max_bounces = 3 (host loop count), anti-aliasing off
    v = dict(DEFAULT, max_bounces=3)
    sa, cam, o, d = _pair(lambda: scenes.sponza_class(0.2), Wd, Hd, v)
    acc, seeds, e, c = o.render(cam, 2, antiAliasing=0)
    d.seed_default()
    d.render(cam, 2, antiAliasing=0)
    assert_bits(d.read_accum(), acc, "3 bounces, no AA")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    d.close()
    # fisheye camera (camera.cl:25-44): sin / cos; rays outside the unit disc are zero rays
    s, view = scenes.cube_scene()
    sa = s.arrays()
    cam = scenes.make_camera(Wd, Hd, view["origin"], view["forward"], fov=60.0, aperture=0.0, type=1)
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    d = Device(Wd, Hd, **DEFAULT)
    d.upload(sa)
    acc, *_ = o.render(cam, 2)
    d.seed_default()
    d.render(cam, 2)
    got = d.read_accum()
    assert_bits(got, acc, "fisheye camera")
    assert np.isfinite(got).all() and got[..., :3].sum() > 0
    d.close()


def test_instance_transform_non_identity_bit_exact():
    """transformRay/transformPosition (tlas.cl:3-8, util.cl:61-87) with a real inverse transform on one instance."""
    Wd, Hd = 128, 72
    from magr_ray_tracer_amd.scenes import Scene, _std_materials, box_tris
    s = Scene()
    _std_materials(s)
    s.AddTriangles(box_tris((-2.4, 0, -0.8), (-0.9, 1.4, 0.6)), "red")
    s.AddQuad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), "grey")
    s.AddQuad((-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1), "white-light")
    s.BuildBLAS(0, 1.0)
    st = s.num_prims
    s.AddTriangles(box_tris((0.5, 0.2, -0.7), (1.9, 1.8, 0.7)), "green")
    s.BuildBLAS(st, 1.0)
    a = np.deg2rad(17.0)
    rot = np.array([[np.cos(a), 0, np.sin(a), 0.13], [0, 1, 0, -0.07], [-np.sin(a), 0, np.cos(a), 0.05], [0, 0, 0, 1]], np.float32)
    s.SetInstanceTransform(1, rot)      # this IS invT (world -> instance), as the reference stores it
    sa = s.arrays()
    assert not np.allclose(sa.blas["invT"][1].reshape(4, 4), np.eye(4))
    view = dict(origin=(0.4, 2.2, 5.5), forward=(0.05, 0.2, 0.97), fov=65.0, aperture=0.02)
    cam = scenes.camera_for(view, Wd, Hd)
    for accel in (0, 1):
        v = dict(DEFAULT, accel=accel)
        o = Oracle(sa, Wd, Hd, **v)
        d = Device(Wd, Hd, **v)
        d.upload(sa)
        acc, seeds, e, c = o.render(cam, 2)
        d.seed_default()
        d.render(cam, 2)
        assert_bits(d.read_accum(), acc, f"rotated instance, accel={accel}")
        _ctr_equal(d.counters(), e, c)
        d.close()


_SPARSE_CACHE = {}


@pytest.mark.parametrize("scene", ["one_blas", "one_blas_bvh4", "two_blas"])
@pytest.mark.parametrize("thin,xcd", [("0", "0"), ("16", "1024"), ("64", "0"), ("4", "64"), ("1", "16384")])
def test_sparse_queues_any_slot_to_lane_map_is_the_same_frame(scene, thin, xcd, monkeypatch):
    """Where a queue slot is traced is free (every slot is traced on its own): the one-ray-per-lane branches keep sparse queues on few XCDs
    (RT355_XCD_RAYS) and on few lanes of every wave (RT355_THIN).  A small frame whose late bounces hold from tens of thousands of rays
    down to a handful, with the two switched off, at their defaults and at extreme values: accumulator, RNG states, work counters and
    the per-pixel `steps` of the last extend must equal the oracle's every time."""
    monkeypatch.setenv("RT355_THIN", thin)
    monkeypatch.setenv("RT355_XCD_RAYS", xcd)
    Wd, Hd = 480, 270
    v = dict(DEFAULT, accel=1 if scene == "one_blas_bvh4" else 0)
    if scene not in _SPARSE_CACHE:
        s, view = scenes.two_blas_scene(0.0, 24) if scene == "two_blas" else scenes.branch_scene()
        sa = s.arrays()
        cam = scenes.camera_for(view, Wd, Hd)
        _SPARSE_CACHE[scene] = (sa, cam) + tuple(Oracle(sa, Wd, Hd, **v).render(cam, 3))
    sa, cam, ref, seeds, e, c = _SPARSE_CACHE[scene]
    d = Device(Wd, Hd, **v)
    d.upload(sa)
    d.seed_default()
    d.enable_steps(True)
    d.render(cam, 3)
    assert_bits(d.read_accum(), ref, f"{scene}: thin {thin}, xcd rays {xcd}")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    d.close()


@pytest.mark.parametrize("accel", [0, 1])
def test_persistent_wavefronts_vs_oracle(accel, monkeypatch):
    """The persistent-wavefront kernels (k_trace_persist / k_trace_persist4) only take over when a queue is longer than one ray per
    resident lane; one workgroup per CU (RT355_TUNE's fifth field) brings that threshold down to 65,536 rays so that a 640x360 frame
    runs every bounce >= 1 and every connect launch through the refill/event machine.  Accumulator, RNG state, counters and the
    per-pixel `steps` heat map must match the oracle bit for bit, and the one-ray-per-lane variant of the same layout."""
    monkeypatch.setenv("RT355_TUNE", "64,20,6,8,1")
    Wd, Hd = 640, 360
    v = dict(DEFAULT, accel=accel)
    s, view = scenes.sponza_class(0.5)
    sa = s.arrays()
    cam = scenes.camera_for(view, Wd, Hd)
    o = Oracle(sa, Wd, Hd, **v)
    ref, seeds, e, c = o.render(cam, 2)
    out = []
    for variant in (0, 2):
        d = Device(Wd, Hd, extend_variant=variant, **v)
        d.upload(sa)
        d.seed_default()
        d.enable_steps(True)
        d.render(cam, 2)
        out.append((d.read_accum(), d.get_seeds(), d.counters(), d.get_steps()))
        d.close()
    assert_bits(out[0][0], ref, "persistent wavefronts vs oracle")
    assert np.array_equal(out[0][1], seeds)
    _ctr_equal(out[0][2], e, c)
    assert out[0][2]["extend_rays"] > 3 * 65536          # the long-queue branch really ran
    assert bits_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][3], out[1][3])


@pytest.mark.parametrize("stack,flat", [("lds", "0,0"), ("spill", "0,0"), ("spill", "1,0"), ("lds", "1,1")])
def test_persistent_wavefronts_through_a_tlas_vs_oracle(stack, flat, monkeypatch):
    """k_trace_persist_tlas (BASELINE config 5's kernel: persistent wavefronts through a multi-BLAS TLAS, TLAS entries on the BLAS stack
    column, the ray transformed once on entering an instance and fetched back from the queue on leaving it).  One workgroup per CU
    brings its long-queue threshold down to 65,536 rays, so a 640x360 frame runs bounces >= 1 and connect through the event loop and
    the later bounces through its per-lane branch.  Two scenes: two SBVH BLAS (glass sphere in one), and three BLAS of which one
    carries a real inverse transform.  Accumulator, RNG state, every work counter (TLAS visits and instance visits included) and the
    per-pixel `steps` must equal the oracle's, and the one-ray-per-lane nested loops' (extend_variant 4)."""
    monkeypatch.setenv("RT355_TUNE", "64,20,6,8,1")
    # "spill": the instantiation for deep trees (config 5's SBVH has 63 levels) - LDS column capped, deeper stack entries in a global
    # per-lane column; a cap of 6 entries makes nearly every ray of these scenes use the global part.  "lds": the whole column in LDS.
    monkeypatch.setenv("RT355_SPILL_CAP" if stack == "spill" else "RT355_NO_SPILL", "6" if stack == "spill" else "1")
    # "e,c": extend / connect through the kernel's event loop (0) or its one-ray-per-lane branch striding over the queue (1; the library's
    # default for multi-BLAS scenes is 1,0)
    monkeypatch.setenv("RT355_TLAS_FLAT", flat)
    Wd, Hd = 960, 540          # bounce 1 still holds more than 65,536 rays
    from magr_ray_tracer_amd.scenes import Scene, _std_materials, box_tris, param_surface

    def three_blas():
        s = Scene()
        _std_materials(s)
        s.AddTriangles(param_surface(lambda U, V: (-2.2 + 1.4 * U, 0.2 + 0.5 * np.sin(6 * U) * np.sin(5 * V) + 0.6, -0.7 + 1.4 * V), 40, 40), "red")
        s.AddQuad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), "grey")
        s.AddQuad((-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1), "white-light")
        s.BuildBLAS(0, 0.0)
        st = s.num_prims
        s.AddTriangles(box_tris((0.5, 0.2, -0.7), (1.9, 1.8, 0.7)), "mirror")
        s.BuildBLAS(st, 1.0)
        st = s.num_prims
        s.AddTriangles(box_tris((-0.4, 0, 1.4), (0.4, 0.8, 2.2)), "sand")
        s.BuildBLAS(st, 1.0)
        a = np.deg2rad(17.0)
        s.SetInstanceTransform(1, np.array([[np.cos(a), 0, np.sin(a), 0.13], [0, 1, 0, -0.07], [-np.sin(a), 0, np.cos(a), 0.05], [0, 0, 0, 1]], np.float32))
        return s, dict(origin=(0.4, 2.2, 5.5), forward=(0.05, 0.2, 0.97), fov=65.0, aperture=0.02)

    for name, fn in (("two SBVH BLAS", lambda: scenes.two_blas_scene(0.0, 48)), ("three BLAS, one rotated", three_blas)):
        s, view = fn()
        sa = s.arrays()
        cam = scenes.camera_for(view, Wd, Hd)
        o = Oracle(sa, Wd, Hd, **DEFAULT)
        ref, seeds, e, c = o.render(cam, 2)
        out = []
        for variant in (0, 4):
            d = Device(Wd, Hd, extend_variant=variant, **DEFAULT)
            d.upload(sa)
            assert d.kernel_info()["persist"] == ((3 if stack == "spill" else 2) if variant == 0 else 0), name
            d.seed_default()
            d.enable_steps(True)
            d.render(cam, 2)
            out.append((d.read_accum(), d.get_seeds(), d.counters(), d.get_steps()))
            d.close()
        assert_bits(out[0][0], ref, f"persistent wavefronts through the TLAS vs oracle ({name})")
        assert np.array_equal(out[0][1], seeds)
        _ctr_equal(out[0][2], e, c)
        assert out[0][2]["extend_tlas_visits"] > 0 and out[0][2]["extend_rays"] > 2 * 65536
        assert bits_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][3], out[1][3]), name
        work = lambda c: {k: v for k, v in c.items() if "issues" not in k and "loop" not in k}
        assert work(out[0][2]) == work(out[1][2]), name      # every work counter, connect's own node / triangle counts included: same any-hit order in both kernels
        if flat == "0,0":
            assert out[0][2]["extend_node_issues"] > 0 and out[0][2]["extend_loop_leaf_events"] > 0      # the event loop really ran (its path-issue counters are kept by the `steps` instantiation, which this test switches on)


def test_persistent_wavefronts_through_a_deeper_tlas_many_instances(monkeypatch):
    """Twelve BLAS on a ring under one TLAS (agglomerative clustering, tlas.cpp:8-52: several TLAS levels, so pending TLAS siblings
    really sit on the tagged stack column between two instance visits, and a ray enters several instances one after the other), two of
    them with a real inverse transform.  k_trace_persist_tlas (event loop for extend and connect, one workgroup per CU so that every
    bounce of the 960x540 frame with more than 65,536 rays runs it) against the oracle - accumulator, RNG state, every work counter -
    and against the nested loops."""
    monkeypatch.setenv("RT355_TUNE", "64,20,6,8,1")
    monkeypatch.setenv("RT355_TLAS_FLAT", "0,0")
    Wd, Hd = 960, 540
    from magr_ray_tracer_amd.scenes import Scene, _std_materials, box_tris, param_surface
    s = Scene()
    _std_materials(s)
    s.AddQuad((-9, 0, -9), (-9, 0, 9), (9, 0, 9), (9, 0, -9), "grey")
    s.AddQuad((-1.5, 6, -1.5), (1.5, 6, -1.5), (1.5, 6, 1.5), (-1.5, 6, 1.5), "white-light")
    s.BuildBLAS(0, 1.0)
    mats = ["red", "green", "sand", "mirror", "white"]
    for k in range(11):
        st = s.num_prims
        a = 2 * np.pi * k / 11
        cx, cz = 3.2 * np.cos(a), 3.2 * np.sin(a)
        if k % 3 == 0:
            s.AddTriangles(param_surface(lambda U, V, cx=cx, cz=cz: (cx - 0.6 + 1.2 * U, 0.3 + 0.9 * V + 0.25 * np.sin(5 * U + k), cz + 0.3 * np.cos(4 * V)), 12, 12), mats[k % 5])
        else:
            s.AddTriangles(box_tris((cx - 0.45, 0.0, cz - 0.45), (cx + 0.45, 0.6 + 0.15 * k, cz + 0.45)), mats[k % 5])
        s.BuildBLAS(st, 1.0 if k % 2 else 0.0)
    for b, ang in ((3, 11.0), (8, -23.0)):
        a = np.deg2rad(ang)
        s.SetInstanceTransform(b, np.array([[np.cos(a), 0, np.sin(a), 0.07], [0, 1, 0, -0.03], [-np.sin(a), 0, np.cos(a), 0.04], [0, 0, 0, 1]], np.float32))
    sa = s.arrays()
    assert len(sa.blas) == 12 and len(sa.tlas) >= 23
    cam = scenes.camera_for(dict(origin=(0.5, 4.2, 8.5), forward=(0.03, 0.42, 0.9), fov=70.0, aperture=0.02), Wd, Hd)
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    ref, seeds, e, c = o.render(cam, 2)
    assert e["tlas_visits"] > 1.5 * e["rays"] and e["inst_visits"] > e["rays"]      # several TLAS levels (depth 5) and more than one instance per ray on average
    out = []
    for variant in (0, 4):
        d = Device(Wd, Hd, extend_variant=variant, **DEFAULT)
        d.upload(sa)
        assert d.kernel_info()["persist"] in ((2, 3) if variant == 0 else (0,))     # (3: stack column capped, its deep end in global memory)
        d.seed_default()
        d.enable_steps(True)
        d.render(cam, 2)
        out.append((d.read_accum(), d.get_seeds(), d.counters(), d.get_steps()))
        d.close()
    assert_bits(out[0][0], ref, "twelve BLAS under a TLAS: persistent wavefronts vs oracle")
    assert np.array_equal(out[0][1], seeds)
    _ctr_equal(out[0][2], e, c)
    assert out[0][2]["extend_node_issues"] > 0
    work = lambda cc: {k: v for k, v in cc.items() if "issues" not in k and "loop" not in k}
    assert bits_equal(out[0][0], out[1][0]) and np.array_equal(out[0][3], out[1][3]) and work(out[0][2]) == work(out[1][2])


def _render_crc(args):
    """Child process: render `frames` frames of the 1280x720 sponza-class scene and return a checksum of accumulator and RNG state."""
    frames, env = args
    import os
    os.environ.update(env)
    import numpy as np
    from magr_ray_tracer_amd import scenes
    from magr_ray_tracer_amd.renderer import Device
    Wd, Hd = 1280, 720
    s, view = scenes.sponza_class(0.5)
    d = Device(Wd, Hd, **DEFAULT)
    d.upload(s.arrays())
    d.seed_default()
    d.render(scenes.camera_for(view, Wd, Hd), frames)
    d.synchronize()
    out = (int(d.read_accum().view(np.uint32).astype(np.uint64).sum()), int(d.get_seeds().astype(np.uint64).sum()))
    d.close()
    return out


def _render_lanes_crc(args):
    """Child process: `lanes` contexts of dist.Lanes render `frames` frames of the 1080p bench scene; checksum of the summed accumulator."""
    lanes, frames = args
    import numpy as np
    from magr_ray_tracer_amd import dist as rdist, scenes
    from magr_ray_tracer_amd.renderer import Device
    from oracle.oracle_py import seed_stream
    Wd, Hd = 1920, 1080
    s, view = scenes.sponza_class(1.0)
    sa = s.arrays()

    def make(m):
        d = Device(Wd, Hd, **DEFAULT)
        d.upload(sa)
        return d
    g = rdist.Lanes(lanes, make, lambda m: seed_stream(m * Wd * Hd, Wd * Hd))
    g.render(scenes.camera_for(view, Wd, Hd), frames)
    g.synchronize()                                   # raises RtError on a device fault
    out = int(g.read_accum().view(np.uint32).astype(np.uint64).sum())
    g.close()
    return out


def test_two_processes_with_two_lanes_each_share_the_gpu():
    """Regression: with one ticket counter per class of workgroups (class = blockIdx mod 32) two processes of two contexts each
    dead-locked k_shade's scan within a few frames - workgroups go to the XCDs round-robin, so an XCD serves only 4 of 32 classes, and
    four partially resident k_shade grids could each hold the XCD another one needed.  With ONE counter any running workgroup draws the
    smallest undrawn tile and nothing depends on which workgroups are resident."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        solo = pool.map(_render_lanes_crc, [(2, 48)])[0]
    with ctx.Pool(2) as pool:
        both = pool.map(_render_lanes_crc, [(2, 48), (2, 48)], chunksize=1)
    assert both == [solo, solo]


def test_shared_gpu_and_oversubscribed_grid():
    """k_shade's ordered scan must not depend on its whole grid being resident: two processes rendering on the same GPU at the same
    time, and a grid of four times what the CUs hold (RT355_SHADE_PER_CU=16), finish without a device fault and reproduce the
    single-process image and RNG state bit for bit."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        solo = pool.map(_render_crc, [(24, {})])[0]
    with ctx.Pool(1) as pool:
        over = pool.map(_render_crc, [(24, {"RT355_SHADE_PER_CU": "16"})])[0]
    with ctx.Pool(2) as pool:
        both = pool.map(_render_crc, [(24, {}), (24, {})], chunksize=1)
    assert over == solo
    assert both[0] == solo and both[1] == solo


@pytest.mark.parametrize("Wd,Hd", [(250, 141), (256, 64), (512, 96), (513, 33)])
def test_odd_sizes_and_full_super_tiles(Wd, Hd):
    """Queue lengths around the scan's structure: 250x141 = 138 tiles (two closed super-tiles and a partial one, width not a multiple
    of the 8x8 primary-ray tiling), 256x64 = exactly one closed super-tile, 512x96 = exactly three, 513x33 ragged in every respect."""
    sa, cam, o, d = _pair(lambda: scenes.sponza_class(0.2), Wd, Hd, DEFAULT)
    acc, seeds, e, c = o.render(cam, 2)
    d.seed_default()
    d.render(cam, 2)
    assert_bits(d.read_accum(), acc, f"{Wd}x{Hd}")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    d.close()


def test_cpp_headless_tick_example(tmp_path):
    """The C++ drop-in path: examples/headless_tick drives Scene / BVH2 / Renderer::Init / Tick / SaveFrame of the host mirror from C++
    (the reference's main loop without its window), with a PNG texture read by Scene::LoadTexture.  Two runs are deterministic,
    BVH2 and BVH4 agree on the image up to the QBVH's tie order, and the PNG carries the frame."""
    import os
    import subprocess
    import zlib
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "headless_tick")
    assert os.path.exists(exe), "examples/headless_tick is built by magr_ray_tracer_amd.build.build_examples()"
    tex = tmp_path / "checker.png"
    yy, xx = np.mgrid[0:64, 0:64]
    img = np.zeros((64, 64, 4), np.float32)
    img[..., 0] = ((xx // 8 + yy // 8) % 2) * 0.8 + 0.1
    img[..., 1] = 0.5
    img[..., 2] = ((xx // 8 + yy // 8 + 1) % 2) * 0.8 + 0.1
    from magr_ray_tracer_amd.scene import save_png
    save_png(tex, img)

    def run(out, *extra):
        r = subprocess.run([exe, "--size", "320", "180", "--spp", "8", "--tex", str(tex), "--out", str(out), *extra],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        line = [l for l in r.stdout.splitlines() if l.startswith("headless_tick:")][0]
        return float(line.split("energy ")[1].split(",")[0]), out.read_bytes()

    e1, p1 = run(tmp_path / "a.png")
    e2, p2 = run(tmp_path / "b.png")
    assert e1 == e2 and p1 == p2 and e1 > 0
    e4, _ = run(tmp_path / "c.png", "--bvh4")
    assert abs(e4 - e1) <= 1e-3 * e1
    # two lanes (two Renderers on one GPU, ticked alternately, accumulators summed): deterministic, same image up to Monte-Carlo noise
    l1, q1 = run(tmp_path / "d.png", "--lanes", "2")
    l2, q2 = run(tmp_path / "e.png", "--lanes", "2")
    assert l1 == l2 and q1 == q2 and abs(l1 - e1) <= 0.03 * e1
    # the PNG decodes to a 320x180 RGB frame that is not black
    assert p1[:8] == b"\x89PNG\r\n\x1a\n" and int.from_bytes(p1[16:20], "big") == 320 and int.from_bytes(p1[20:24], "big") == 180
    idat = b"".join(p1[i + 8:i + 8 + int.from_bytes(p1[i:i + 4], "big")] for i in _png_chunks(p1) if p1[i + 4:i + 8] == b"IDAT")
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(180, 1 + 320 * 3)
    assert raw[:, 1:].mean() > 10


def _png_chunks(b):
    i = 8
    while i + 12 <= len(b):
        yield i
        i += 12 + int.from_bytes(b[i:i + 4], "big")


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_random_triangle_soups(seed, big=False):
    """Random triangle soups (slivers, overlapping and coplanar triangles, mirrors, several lights of different size), random camera,
    random kernel variant and SBVH alpha, random frame size: accumulator, RNG state and work counters bit-exact against the oracle."""
    from magr_ray_tracer_amd.scene import Scene, material
    rng = np.random.default_rng(1000 + seed)
    s = Scene()
    s.AddMaterial("a", material(color=rng.random(3)))
    s.AddMaterial("b", material(color=rng.random(3)))
    s.AddMaterial("m", material(color=rng.random(3), specular=float(rng.choice([0.3, 0.9, 1.0]))))
    s.AddMaterial("l1", material(color=(1, 1, 1), light=True, emittance=tuple(rng.random(3) * 40 + 5)))
    s.AddMaterial("l2", material(color=(1, 1, 1), light=True, emittance=tuple(rng.random(3) * 10 + 1)))
    n = int(rng.integers(40, 400))
    c = rng.random((n, 1, 3)) * 8 - 4
    size = np.where(rng.random((n, 1, 1)) < 0.15, 3.0, 0.6)
    tris = (c + (rng.random((n, 3, 3)) - 0.5) * size).astype(np.float32)
    tris[: n // 10, 2] = tris[: n // 10, 0] + (tris[: n // 10, 1] - tris[: n // 10, 0]) * 1.0001 + 1e-4      # slivers
    if n > 60:
        tris[50:55] = tris[45:50]                                                                              # exact duplicates (ties)
    names = rng.choice(["a", "b", "m"], size=n, p=[0.45, 0.35, 0.2])
    for k in ("a", "b", "m"):
        sel = tris[names == k]
        if len(sel):
            s.AddTriangles(sel, k)
    s.AddTriangles(np.array([[[-6, 7, -6], [6, 7, -6], [6, 7, 6]], [[6, 7, 6], [-6, 7, 6], [-6, 7, -6]]], np.float32), "l1", flipNormal=True)
    s.AddTriangles((rng.random((2, 3, 3)) * 2 + np.array([2.0, 1.0, -3.0])).astype(np.float32), "l2")
    s.AddTriangles(np.array([[[-9, -4.5, -9], [9, -4.5, 9], [9, -4.5, -9]], [[-9, -4.5, -9], [-9, -4.5, 9], [9, -4.5, 9]]], np.float32), "a")
    s.BuildBLAS(0, alpha=float(rng.choice([1.0, 1e-5, 0.0])))
    sa = s.arrays()
    Wd, Hd = int(rng.integers(17, 200)), int(rng.integers(9, 120))
    if big:   # tools/deep_fuzz.py: long queues (persistent-wavefront branch with RT355_TUNE=...,1; several super-tiles in k_shade)
        Wd, Hd = int(rng.integers(300, 520)), int(rng.integers(200, 300))
    v = dict(DEFAULT, accel=int(rng.integers(0, 2)), shading=int(rng.integers(0, 2)), sampling=int(rng.integers(0, 2)),
             russian_roulette=bool(rng.integers(0, 2)), filter_fireflies=bool(rng.integers(0, 2)))
    org = rng.random(3) * 6 - 3 + np.array([0, 0, 9.0])
    cam = scenes.make_camera(Wd, Hd, tuple(org), tuple(np.array([0.0, 0.1, 1.0]) + (rng.random(3) - 0.5) * 0.4), fov=float(rng.integers(40, 120)),
                             aperture=float(rng.choice([0.0, 0.1])))
    o = Oracle(sa, Wd, Hd, **v)
    d = Device(Wd, Hd, **v)
    d.upload(sa)
    frames = 3
    acc, seeds, e, c = o.render(cam, frames)
    d.seed_default()
    d.render(cam, frames)
    assert_bits(d.read_accum(), acc, f"seed {seed}: {n} tris {Wd}x{Hd} {v}")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    d.close()


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_random_multi_blas_soups(seed, monkeypatch, big=False):
    """Random triangle soups cut into 2-6 BLAS (each its own BuildBLAS with its own SBVH alpha) under a TLAS, some instances with a
    random rigid inverse transform, random camera / kernel variant / frame size, and a random way through k_trace_persist_tlas: LDS
    stack or a stack capped at 6-8 entries with the global spill, extend and connect through the event loop or the one-ray-per-lane
    branch (one workgroup per CU, so queues above 65,536 rays take the long-queue code).  Accumulator, RNG state and every work counter
    (TLAS and instance visits included) bit-exact against the oracle."""
    from magr_ray_tracer_amd.scene import Scene, material
    rng = np.random.default_rng(7000 + seed)
    monkeypatch.setenv("RT355_TUNE", "64,20,6,8,1")
    monkeypatch.setenv("RT355_TLAS_FLAT", str(rng.choice(["0,0", "1,0", "1,1", "0,1"])))
    cap = int(rng.choice([0, 6, 8]))
    if cap:
        monkeypatch.setenv("RT355_SPILL_CAP", str(cap))
    else:
        monkeypatch.setenv("RT355_NO_SPILL", "1")
    s = Scene()
    s.AddMaterial("a", material(color=rng.random(3)))
    s.AddMaterial("b", material(color=rng.random(3)))
    s.AddMaterial("m", material(color=rng.random(3), specular=float(rng.choice([0.3, 0.9, 1.0]))))
    s.AddMaterial("g", material(color=(1, 1, 1), dielectric=True, n1=1.0, n2=1.3, specular=0.05, absorption=(0.02, 0.05, 0.01)))
    s.AddMaterial("l1", material(color=(1, 1, 1), light=True, emittance=tuple(rng.random(3) * 40 + 5)))
    nb = int(rng.integers(2, 7))
    # BLAS 0: floor + ceiling light (so that every ray has something to hit and NEE has a light)
    s.AddTriangles(np.array([[[-9, -4.5, -9], [9, -4.5, 9], [9, -4.5, -9]], [[-9, -4.5, -9], [-9, -4.5, 9], [9, -4.5, 9]]], np.float32), "a")
    s.AddTriangles(np.array([[[-6, 7, -6], [6, 7, -6], [6, 7, 6]], [[6, 7, 6], [-6, 7, 6], [-6, 7, -6]]], np.float32), "l1", flipNormal=True)
    s.BuildBLAS(0, alpha=1.0)
    for b in range(1, nb):
        st = s.num_prims
        n = int(rng.integers(20, 160))
        centre = rng.random(3) * 6 - 3
        c = centre + (rng.random((n, 1, 3)) - 0.5) * 3.0
        size = np.where(rng.random((n, 1, 1)) < 0.15, 2.0, 0.5)
        tris = (c + (rng.random((n, 3, 3)) - 0.5) * size).astype(np.float32)
        if n > 30:
            tris[20:24] = tris[16:20]                                     # exact duplicates (ties)
        names = rng.choice(["a", "b", "m", "g"], size=n, p=[0.4, 0.3, 0.2, 0.1])
        for k in ("a", "b", "m", "g"):
            sel = tris[names == k]
            if len(sel):
                s.AddTriangles(sel, k)
        s.BuildBLAS(st, alpha=float(rng.choice([1.0, 1e-5, 0.0])))
        if rng.random() < 0.5:                                            # a rigid world -> instance transform (rotation about y + a small shift)
            a = float(rng.random() * 0.8 - 0.4)
            t = (rng.random(3) - 0.5) * 0.6
            s.SetInstanceTransform(b, np.array([[np.cos(a), 0, np.sin(a), t[0]], [0, 1, 0, t[1]], [-np.sin(a), 0, np.cos(a), t[2]], [0, 0, 0, 1]], np.float32))
    sa = s.arrays()
    assert len(sa.blas) == nb
    Wd, Hd = int(rng.integers(17, 200)), int(rng.integers(9, 120))
    if big or seed % 4 == 0:        # every fourth seed (and tools/deep_fuzz.py multi big): queues long enough for the event loop and the refill machinery
        Wd, Hd = int(rng.integers(300, 420)), int(rng.integers(200, 260))
    v = dict(DEFAULT, accel=int(rng.random() < 0.2), shading=int(rng.integers(0, 2)), sampling=int(rng.integers(0, 2)),
             russian_roulette=bool(rng.integers(0, 2)), filter_fireflies=bool(rng.integers(0, 2)))
    org = rng.random(3) * 6 - 3 + np.array([0, 0, 9.0])
    cam = scenes.make_camera(Wd, Hd, tuple(org), tuple(np.array([0.0, 0.1, 1.0]) + (rng.random(3) - 0.5) * 0.4), fov=float(rng.integers(40, 120)),
                             aperture=float(rng.choice([0.0, 0.1])))
    o = Oracle(sa, Wd, Hd, **v)
    d = Device(Wd, Hd, **v)
    d.upload(sa)
    if v["accel"] == 0:
        assert d.kernel_info()["persist"] in (2, 3)
    frames = 3
    acc, seeds, e, c = o.render(cam, frames)
    d.seed_default()
    d.render(cam, frames)
    assert_bits(d.read_accum(), acc, f"seed {seed}: {nb} BLAS {Wd}x{Hd} {v} cap {cap}")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    d.close()


def test_textured_plane_and_stray_texture_indices():
    """A textured plane seen on both sides of its origin (negative u and v: the reference's lookup then lands up to a whole texture past
    the texture's window, primitives.cl:137-146) with the texture LAST in the atlas, and a triangle whose uv reach exactly 1: HIP and
    oracle agree bit for bit, texels outside the atlas read as zero in both (the reference reads out of bounds there)."""
    from magr_ray_tracer_amd.scene import Scene, material
    s = Scene()
    s.AddMaterial("white", material(color=(0.8, 0.8, 0.8)))
    s.AddMaterial("light", material(color=(1, 1, 1), light=True, emittance=(30, 30, 30)))
    rng = np.random.default_rng(2)
    t1 = np.zeros((8, 8, 4), np.float32); t1[..., :3] = rng.random((8, 8, 3))
    t2 = np.zeros((4, 16, 4), np.float32); t2[..., :3] = rng.random((4, 16, 3))
    s.AddTexture("first", t1)
    s.AddTexture("last", t2)
    s.AddPlane((0, 1, 0), 0.0, "last")
    s.AddTriangle((-3, 0.5, -2), (3, 0.5, -2), (0, 3, -2), "first", uv0=(0, 0), uv1=(1, 0), uv2=(1, 1))
    s.AddQuad((-2, 6, -2), (2, 6, -2), (2, 6, 2), (-2, 6, 2), "light", flipNormal=True)
    s.BuildBLAS(0, 1.0)
    sa = s.arrays()
    Wd, Hd = 160, 90
    cam = scenes.make_camera(Wd, Hd, (0.3, 2.0, 6.0), (0.0, 0.25, 1.0), fov=80.0, aperture=0.0)
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    d = Device(Wd, Hd, **DEFAULT)
    d.upload(sa)
    acc, seeds, e, c = o.render(cam, 4)
    d.seed_default()
    d.render(cam, 4)
    assert_bits(d.read_accum(), acc, "textured plane")
    assert np.array_equal(d.get_seeds(), seeds)
    assert acc[..., :3].sum() > 0
    d.close()


def test_lanes_interleaved_sample_streams_match_oracle():
    """dist.Lanes: two contexts on one GPU render disjoint sample streams of the same frame with their launches interleaved (bench.py's
    default).  Each lane equals the oracle run on its seed slice, the rank's image is the sum of the lanes in lane order - bit for
    bit - and it differs from the single-stream image only in which random numbers were drawn."""
    from magr_ray_tracer_amd import dist as rdist
    Wd, Hd, frames, lanes = 160, 90, 5, 2
    s, view = scenes.sponza_class(0.2)
    sa = s.arrays()
    cam = scenes.camera_for(view, Wd, Hd)

    def make(m):
        d = Device(Wd, Hd, **DEFAULT)
        d.upload(sa)
        return d

    def seeds_for(m):
        p = rdist.plan("samples", Wd, Hd, 0, 1, m, lanes)
        return seed_stream(p["seed_first"], p["seed_count"])

    g = rdist.Lanes(lanes, make, seeds_for)
    g.render(cam, frames)
    g.synchronize()
    total = g.read_accum()
    parts = rdist.lane_frames(frames, lanes)
    assert parts == [3, 2]
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    ref = None
    for m in range(lanes):
        acc, seeds, e, c = o.render(cam, parts[m], seeds=seeds_for(m))
        assert_bits(g.devs[m].read_accum(), acc, f"lane {m}")
        assert np.array_equal(g.devs[m].get_seeds(), seeds)
        ref = acc if ref is None else ref + acc
    assert_bits(total, ref, "sum of lanes")
    one = Device(Wd, Hd, **DEFAULT)
    one.upload(sa)
    one.seed_default()
    one.render(cam, frames)
    single = one.read_accum()
    assert not bits_equal(single, total) and abs(float(single[..., :3].sum()) / float(total[..., :3].sum()) - 1) < 0.05
    one.close()
    g.close()


# ---- round 3: lanes behind one handle (rt_group_*) ----------------------------------------------------------------------------------
def test_group_of_lanes_matches_oracle_sample_streams_and_a_single_lane_is_the_plain_renderer():
    """rt_group_* (include/rt355.h): one accumulation as `lanes` interleaved sample streams behind ONE handle - own context, stream and
    seed slice per lane, one device copy of the scene, frames dealt round-robin, accumulator = the lanes added up in lane order on the
    device.  Three lanes, seven frames (3 + 2 + 2): every lane's accumulator and the group's sum equal the oracle rendering the same seed
    slices, bit for bit; post-processing works on the summed accumulator with the group's frame count; a group of ONE lane is the plain
    single-context renderer bit for bit; the streams of the group were measured to run concurrently."""
    from magr_ray_tracer_amd import dist as rdist
    from magr_ray_tracer_amd.renderer import Group
    Wd, Hd, frames, lanes, first = 160, 90, 7, 3, 2
    s, view = scenes.sponza_class(0.2)
    sa = s.arrays()
    cam = scenes.camera_for(view, Wd, Hd)
    g = Group(Wd, Hd, lanes=lanes, **DEFAULT)
    g.upload(sa)
    assert g.concurrency() == lanes and len(g) == lanes
    g.seed(first)                                     # lane m renders sample stream first + m
    cam["focalLength"] = g.focus(Wd // 2, Hd // 2, cam)
    g.render(cam, 4)
    g.render(cam, 3)                                  # the round-robin continues where the first call stopped
    total = g.read_accum()
    assert g.frames() == frames
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    exp = None
    for m, n in enumerate(rdist.lane_frames(frames, lanes)):
        acc, _, _, _ = o.render(cam, n, seeds=seed_stream((first + m) * Wd * Hd, Wd * Hd))
        assert_bits(g.devs[m].read_accum(), acc, f"lane {m}")
        exp = acc if exp is None else exp + acc
    assert_bits(total, exp, "group accumulator = lanes added in lane order")
    img, rgba = g.postproc()
    from oracle.oracle_py import postproc as orc_postproc
    ref_img, _ = orc_postproc(exp, frames, 0.0, 0.9, 0.0)
    assert np.abs(img[..., :3] - ref_img[..., :3]).max() < 3e-6 and rgba.shape == (Hd, Wd, 4)
    g.reset()
    assert g.frames() == 0 and not g.read_accum().any()
    g.close()
    one = Group(Wd, Hd, lanes=1, **DEFAULT)
    one.upload(sa)
    one.seed(0)
    one.render(cam, 3)
    d = Device(Wd, Hd, **DEFAULT)
    d.upload(sa)
    d.seed_default()
    d.render(cam, 3)
    assert_bits(one.read_accum(), d.read_accum(), "a group of one lane vs the plain context")
    assert one.devs[0].kernel_info() == d.kernel_info()       # same kernels, same grids: nothing of the sharing footprint
    one.close()
    d.close()


def test_renderer_mirror_with_lanes_ticks_whole_rounds():
    """The C++ Renderer mirror with `lanes` = 2 (host/renderer.cpp): a Tick() adds two frames - one per lane - to ONE accumulation,
    settings->frames counts them, and the accumulator it reads back is the sum of the two sample streams as the oracle renders them."""
    from magr_ray_tracer_amd.renderer import Renderer
    Wd, Hd = 160, 90
    s, view = scenes.cube_scene()
    sa = s.arrays()
    r = Renderer(s, Wd, Hd)
    r.SetLanes(2)
    r.SetCamera(view["origin"], view["forward"], view["fov"], view["aperture"])
    r.Init()
    r.Tick(3)
    assert r.frames() == 1 + 3 * 2
    got, energy = r.read()
    cam = r.camera()
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    exp = None
    for m in range(2):
        acc, _, _, _ = o.render(cam, 3, seeds=seed_stream(m * Wd * Hd, Wd * Hd))
        exp = acc if exp is None else exp + acc
    assert_bits(got, exp, "Renderer with two lanes")
    assert energy > 0
    r.close()


# ---- round 2 -----------------------------------------------------------------------------------------------------------------
def test_bench_starts_its_own_ranks_two_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` from a bare shell (no torch.distributed.run): the parent spawns the two ranks itself.  Rehearsal
    form for a 1-GPU box (--same-device --backend gloo; RCCL refuses two ranks on one device).  The reduced accumulator must equal
    the sum of the two ranks' sample streams rendered one after the other in this process, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    Wd, Hd, steps = 640, 360, 4
    dump = tmp_path / "acc.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--steps", str(steps),
           "--warmup", "0", "--lanes", "1", "--no-cpu-baseline", "--no-repeat", "--width", str(Wd), "--height", str(Hd), "--detail", "0.2",
           "--dump-accum", str(dump)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == steps and line["repeats"] == 1
    assert len(line["rank_s"]["render"]) == 2 and len(line["rank_s"]["all_reduce"]) == 2 and line["value"] > 0
    roof = line["roofline"]
    assert roof["kernel"].startswith("extend = k_trace_persist<false>") and roof["algorithmic"]["gbs"] > 0 and roof["avg_launch_ms"] > 0
    # the kernel's own launch time comes from ONE context with the GPU to itself: its seven extend launches fit inside that context's frame
    assert 7 * roof["avg_launch_ms"] <= 1e3 * Wd * Hd * 2 / (line["value_single_context"] * 1e6) * 1.02
    assert all(0 < v["frac"] <= 1 for v in roof["levels"].values())     # (empty unless a committed PMC measurement matches this frame size)
    got = np.load(dump)
    s, view = scenes.sponza_class(0.2)
    sa = s.arrays(bvh4=False)
    cam = scenes.camera_for(view, Wd, Hd)
    exp = np.zeros((Hd, Wd, 4), np.float32)
    for rank in range(2):
        d = Device(Wd, Hd, **DEFAULT)
        d.upload(sa)
        if rank == 0:
            cam["focalLength"] = d.focus(Wd // 2, Hd // 2, cam)
        d.set_seeds(seed_stream(rank * Wd * Hd, Wd * Hd))
        d.render(cam, steps)
        exp = exp + d.read_accum()
        d.close()
    assert_bits(got, exp, "2-rank reduced accumulator vs the two sample streams")
    # a failing rank must surface as a non-zero exit code of the launcher
    bad = subprocess.run(cmd[:-2] + ["--shard", "nonsense"], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert bad.returncode != 0


def test_bench_fixed_image_split_over_two_ranks_with_lanes_inside_bands(tmp_path):
    """`bench.py --total-steps T` renders a FIXED T-spp image split over the ranks ("scaling": "strong"; the north star's 256 / 1024 spp
    over 8 GPUs) - here config 4's plan in miniature: two ranks (rehearsal form: same device, gloo), the frame cut into row bands dealt
    round-robin (ibands), TWO lanes per band (sample streams 0 and 1 of those rows, rt_group handles), 5 spp in all.  The reduced
    accumulator equals the oracle rendering every (band, lane) with its seed slice and frame share, bit for bit: a pixel has one non-zero
    addend across ranks, and a band's lanes are added in lane order."""
    import json
    import os
    import subprocess
    import sys
    from magr_ray_tracer_amd import dist as rdist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    Wd, Hd, total, lanes, rows = 320, 180, 5, 2, 23
    dump = tmp_path / "acc.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--total-steps", str(total),
           "--shard", "ibands", "--band-rows", str(rows), "--lanes", str(lanes), "--warmup", "0", "--no-cpu-baseline", "--no-repeat",
           "--width", str(Wd), "--height", str(Hd), "--detail", "0.2", "--dump-accum", str(dump)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == total and line["config"]["lanes"] == lanes
    assert line["value"] > 0 and abs(line["value"] - Wd * Hd * total / (line["ms_per_step"] * total * 1e-3) / 1e6) < 1e-2 * line["value"]
    got = np.load(dump)
    s, view = scenes.sponza_class(0.2)
    sa = s.arrays(bvh4=False)
    cam = scenes.camera_for(view, Wd, Hd)
    d = Device(Wd, Hd, y0=0, y1=rows, **DEFAULT)             # bench focuses through rank 0's first context (band 0)
    d.upload(sa)
    cam["focalLength"] = d.focus(Wd // 2, Hd // 2, cam)
    d.close()
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    exp = np.zeros((Hd, Wd, 4), np.float32)
    for rank in range(2):
        for p0 in rdist.plans("ibands", Wd, Hd, rank, 2, band_rows=rows):
            band = None
            for m, frames in enumerate(rdist.lane_frames(total, lanes)):      # [3, 2]
                p = rdist.plans("ibands", Wd, Hd, rank, 2, m, lanes, band_rows=rows)[[q["y0"] for q in rdist.plans("ibands", Wd, Hd, rank, 2, band_rows=rows)].index(p0["y0"])]
                acc = np.zeros((Hd, Wd, 4), np.float32)
                o.render(cam, frames, accum=acc, seeds=seed_stream(p["seed_first"], p["seed_count"]), y0=p["y0"], y1=p["y1"])
                band = acc if band is None else band + acc
            exp = exp + band
    assert_bits(got, exp, "fixed 5-spp image, two ranks x interleaved bands x two lanes")


def test_bench_scene_1080p_band_vs_oracle_counters_equal():
    """The bench workload itself - sponza_class(1.0), 264,946 triangles, tree depth 21 (its own LDS stack size, occupancy and
    persistent grid), 1920x1080 - against the oracle on a 16-row band: accumulator and RNG state bit for bit, extend work counters
    equal (connect runs its own any-hit order: ray count equal)."""
    Wd, Hd, y0, y1 = 1920, 1080, 532, 548
    s, view = scenes.sponza_class(1.0)
    sa = s.arrays(bvh4=False)
    cam = scenes.camera_for(view, Wd, Hd)
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    d = Device(Wd, Hd, y0=y0, y1=y1, **DEFAULT)
    d.upload(sa)
    info = d.kernel_info()
    assert info["persist"] == 1 and info["layout"] == 1 and info["stack_entries"] >= 22
    cam["focalLength"] = d.focus(Wd // 2, Hd // 2, cam)
    ref, seeds, e, c = o.render(cam, 2, y0=y0, y1=y1)
    d.seed_default()
    d.render(cam, 2)
    assert_bits(d.read_accum(), ref, "bench scene band vs oracle")
    assert np.array_equal(d.get_seeds(), seeds)
    dc = d.counters()
    for k in ("rays", "tlas_visits", "inst_visits", "node_visits", "prim_tests"):
        assert dc["extend_" + k] == e[k], (k, dc["extend_" + k], e[k])
    assert dc["connect_rays"] == c["rays"]
    d.close()


def test_8k_full_frame_vs_oracle_bit_exact():
    """Maximum size: a full 7680x4320 frame (33.2 M pixels, 232 M shadow-queue slots, queues of 16 B x 33 M and more) against the oracle:
    accumulator, RNG state and extend work counters.  Exercises the persistent traversal over queues sixteen times the bench frame's,
    the ordered scan of k_shade over 64,800 tiles and every index that must not wrap at 2^31 bytes."""
    Wd, Hd = 7680, 4320
    s, view = scenes.sponza_class(0.2)
    sa = s.arrays(bvh4=False)
    cam = scenes.camera_for(view, Wd, Hd)
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    d = Device(Wd, Hd, **DEFAULT)
    d.upload(sa)
    cam["focalLength"] = d.focus(Wd // 2, Hd // 2, cam)
    ref, seeds, e, c = o.render(cam, 1)
    d.seed_default()
    d.render(cam, 1)
    assert_bits(d.read_accum(), ref, "8K frame vs oracle")
    assert np.array_equal(d.get_seeds(), seeds)
    dc = d.counters()
    for k in ("rays", "tlas_visits", "inst_visits", "node_visits", "prim_tests"):
        assert dc["extend_" + k] == e[k], (k, dc["extend_" + k], e[k])
    assert dc["connect_rays"] == c["rays"]
    d.close()


def test_stage_entry_points_respect_the_contexts_max_bounces():
    """Shadow queue and counter rows are sized by cfg.max_bounces: a stage call beyond it must be refused, not written past them."""
    sa, cam, o, d = _pair(scenes.cube_scene, 64, 36, dict(DEFAULT, max_bounces=2))
    d.seed_default()
    d.render(cam, 1)
    for fn, args in ((d.stage_shade, (3,)), (d.stage_shade, (2,)), (d.stage_connect, (0, 6)), (d.stage_connect, (2, 2)), (d.stage_extend, (3,)),
                     (d.get_shadow, (0, 2))):
        with pytest.raises(RtError):
            fn(*args)
    d.stage_connect(0, 1)
    d.close()


def test_upload_rejects_tlas_cycles_and_overdeep_tlas():
    s, view = scenes.two_blas_scene(alpha=1.0, n=6)
    sa = s.arrays()
    d = Device(32, 18, **DEFAULT)
    d.upload(sa)
    import copy
    bad = copy.copy(sa)
    bad.tlas = sa.tlas.copy()
    root = int(np.nonzero(bad.tlas["leftRight"] != 0)[0][0])
    bad.tlas["leftRight"][root] = (root << 16) | (int(bad.tlas["leftRight"][root]) & 0xffff)     # right child = the node itself
    with pytest.raises(RtError, match="reachable twice|cycle"):
        d.upload(bad)
    # a 40-deep chain of interior nodes: deeper than the 32-entry TLAS stack of traverse_tlas (tlas.cl:42)
    deep = copy.copy(sa)
    n = 41
    t = np.zeros(2 * n + 1, dtype=sa.tlas.dtype)
    for i in range(n):                       # node i: left = leaf n + i, right = node i + 1 (the last one: leaf 2n)
        t["leftRight"][i] = (((i + 1) if i < n - 1 else 2 * n) << 16) | (n + i)
        t["aabbMin"][i], t["aabbMax"][i] = sa.tlas["aabbMin"][0], sa.tlas["aabbMax"][0]
    for i in range(n, 2 * n + 1):
        t["leftRight"][i], t["BLASidx"][i] = 0, i % 2
        t["aabbMin"][i], t["aabbMax"][i] = sa.tlas["aabbMin"][0], sa.tlas["aabbMax"][0]
    deep.tlas = t
    with pytest.raises(RtError, match="depth"):
        d.upload(deep)
    d.upload(sa)           # the context is still usable
    d.close()


def test_obj_mtl_png_model_renders_textured(tmp_path):
    """LoadModel("m.obj", "white") with an MTL that names a PNG: the file-loaded texture is what the hit primitives are shaded with
    (HIP vs oracle bit for bit, and the image differs from the same model forced to the default material)."""
    from test_io_postproc_cpu import write_textured_obj
    from magr_ray_tracer_amd.scene import Scene, material
    write_textured_obj(tmp_path)

    def build(force):
        s = Scene()
        s.AddMaterial("white", material(color=(.8, .8, .8)))
        s.AddMaterial("lamp", material(color=(1, 1, 1), light=True, emittance=(30, 30, 30)))
        s.LoadModel(tmp_path / "m.obj", "white", forceDefaultMat=force)
        s.AddQuad((-1, -1, 2.5), (2, -1, 2.5), (2, 2, 2.5), (-1, 2, 2.5), "lamp")     # faces -z, towards the quad
        s.BuildBLAS(0, 1.0)
        return s.arrays()
    Wd, Hd = 96, 54
    view = dict(origin=(0.5, 0.5, -1.6), forward=(0, 0, -1), fov=60.0, aperture=0.0)
    cam = scenes.camera_for(view, Wd, Hd)
    imgs = []
    for force in (False, True):
        sa = build(force)
        o = Oracle(sa, Wd, Hd, **DEFAULT)
        d = Device(Wd, Hd, **DEFAULT)
        d.upload(sa)
        ref, seeds, e, c = o.render(cam, 3)
        d.seed_default()
        d.render(cam, 3)
        got = d.read_accum()
        assert_bits(got, ref, "OBJ+MTL+PNG scene")
        imgs.append(got)
        if not force:   # primary rays do land on the textured triangles
            rays = o.generate(cam, 0, Wd * Hd, seed_stream(0, Wd * Hd))
            o.extend(rays)
            hit = rays["primIdx"][rays["primIdx"] >= 0]
            assert (sa.mats["texIdx"][sa.prims["matIdx"][hit]] != -1).sum() > 200
        d.close()
    assert np.abs(imgs[0] - imgs[1]).max() > 0.05


@pytest.mark.parametrize("case", ["mixed", "branch", "branch_kajiya_hemi", "two_blas_glass", "fisheye", "config5_small"])
def test_scenes_with_transcendentals_are_bit_exact(case):
    """Glass (exp), sphere lights and the fisheye camera (sin / cos), textured spheres (acos / atan2): oracle and HIP path evaluate
    the same Cephes-style sequences of IEEE operations (rt355_kernels.h rt_expf ...; oracle.c orc_expf ...), so these scenes are
    held to the same standard as triangle scenes - accumulator, RNG state and extend counters bit for bit."""
    v = dict(DEFAULT)
    Wd, Hd, frames = 192, 108, 3
    if case == "mixed":
        s, view = scenes.mixed_scene()
    elif case.startswith("branch"):
        s, view = scenes.branch_scene()
        if case == "branch_kajiya_hemi":
            v = dict(DEFAULT, shading=0, sampling=0, russian_roulette=False)
    elif case == "two_blas_glass":
        s, view = scenes.two_blas_scene(alpha=0.0, n=16)
    elif case == "fisheye":
        s, view = scenes.branch_scene()
        view = dict(view, type=1, fov=75.0)
    else:
        s, view = scenes.config5_scene(alpha=1.0, decimate=4)
    sa = s.arrays()
    cam = scenes.camera_for(view, Wd, Hd)
    o = Oracle(sa, Wd, Hd, **v)
    d = Device(Wd, Hd, **v)
    d.upload(sa)
    f = o.focus(Wd // 2, Hd // 2, cam)
    assert d.focus(Wd // 2, Hd // 2, cam) == f
    cam["focalLength"] = f
    ref, seeds, e, c = o.render(cam, frames)
    d.seed_default()
    d.render(cam, frames)
    assert_bits(d.read_accum(), ref, case + " accumulator")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    d.close()


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_random_mixed_scenes(seed, big=False):
    """Random scenes with everything the path supports at once - triangle soups, diffuse / mirror / glass / textured / emissive spheres,
    textured triangles with wrapping uv, glass triangles with absorption, triangle and sphere lights, pinhole or fisheye camera, every
    kernel variant, SBVH - bit-exact against the oracle (accumulator, RNG state, extend counters).  Possible since exp / sin / cos /
    acos / atan2 are the same IEEE sequences on both sides."""
    from magr_ray_tracer_amd.scene import Scene, material
    rng = np.random.default_rng(77000 + seed)
    s = Scene()
    s.AddMaterial("a", material(color=rng.random(3)))
    s.AddMaterial("b", material(color=rng.random(3) * 0.9 + 0.1))
    s.AddMaterial("m", material(color=rng.random(3), specular=float(rng.choice([0.3, 0.9, 1.0]))))
    s.AddMaterial("g", material(color=(1, 1, 1), dielectric=True, n1=1.0, n2=float(rng.choice([1.1, 1.33, 1.5, 2.4])), specular=float(rng.choice([0.0, 0.04])),
                                absorption=tuple(rng.random(3) * float(rng.choice([0.0, 0.2, 2.0])))))
    s.AddMaterial("l1", material(color=(1, 1, 1), light=True, emittance=tuple(rng.random(3) * 40 + 5)))
    s.AddMaterial("l2", material(color=(1, 1, 1), light=True, emittance=tuple(rng.random(3) * 10 + 1)))
    tw, th = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    tex = np.zeros((th, tw, 4), np.float32)
    tex[..., :3] = rng.random((th, tw, 3))
    s.AddTexture("t", tex)
    n = int(rng.integers(20, 200))
    c = rng.random((n, 1, 3)) * 8 - 4
    size = np.where(rng.random((n, 1, 1)) < 0.2, 3.0, 0.8)
    tris = (c + (rng.random((n, 3, 3)) - 0.5) * size).astype(np.float32)
    names = rng.choice(["a", "b", "m", "g", "t"], size=n, p=[0.3, 0.25, 0.15, 0.15, 0.15])
    for k in ("a", "b", "m", "g"):
        sel = tris[names == k]
        if len(sel):
            s.AddTriangles(sel, k)
    sel = tris[names == "t"]
    if len(sel):
        s.AddTriangles(sel, "t", uvs=(rng.random((len(sel), 3, 2)) * 5 - 2).astype(np.float32))
    for _ in range(int(rng.integers(2, 9))):
        s.AddSphere(tuple(rng.random(3) * 7 - 3.5), float(rng.random() * 1.2 + 0.15), str(rng.choice(["a", "m", "g", "t", "l2"])))
    s.AddTriangles(np.array([[[-6, 7, -6], [6, 7, -6], [6, 7, 6]], [[6, 7, 6], [-6, 7, 6], [-6, 7, -6]]], np.float32), "l1", flipNormal=True)
    s.AddTriangles(np.array([[[-9, -4.5, -9], [9, -4.5, 9], [9, -4.5, -9]], [[-9, -4.5, -9], [-9, -4.5, 9], [9, -4.5, 9]]], np.float32), "b")
    s.BuildBLAS(0, alpha=float(rng.choice([1.0, 1.0, 0.0])))
    sa = s.arrays()
    Wd, Hd = int(rng.integers(17, 200)), int(rng.integers(9, 120))
    if big:
        Wd, Hd = int(rng.integers(300, 520)), int(rng.integers(200, 300))
    v = dict(DEFAULT, accel=int(rng.integers(0, 2)), shading=int(rng.integers(0, 2)), sampling=int(rng.integers(0, 2)),
             russian_roulette=bool(rng.integers(0, 2)), filter_fireflies=bool(rng.integers(0, 2)))
    org = rng.random(3) * 6 - 3 + np.array([0, 0, 9.0])
    cam = scenes.make_camera(Wd, Hd, tuple(org), tuple(np.array([0.0, 0.1, 1.0]) + (rng.random(3) - 0.5) * 0.4), fov=float(rng.integers(40, 120)),
                             aperture=float(rng.choice([0.0, 0.1])), type=int(rng.random() < 0.25))
    o = Oracle(sa, Wd, Hd, **v)
    d = Device(Wd, Hd, **v)
    d.upload(sa)
    frames = 3
    acc, seeds, e, c = o.render(cam, frames)
    d.seed_default()
    d.render(cam, frames)
    assert_bits(d.read_accum(), acc, f"seed {seed}: {n} tris {Wd}x{Hd} {v}")
    assert np.array_equal(d.get_seeds(), seeds)
    _ctr_equal(d.counters(), e, c)
    d.close()


def test_interleaved_bands_on_one_gpu_match_oracle_bands():
    """The interleaved-band plan (dist.plans("ibands")): each rank runs one context per owned band, all rendering into the rank's
    full-frame accumulator; the sum over ranks of those accumulators equals the oracle rendering the same bands (each with the frame's own
    seed slice), bit for bit.  Two "ranks" are played one after the other on the one GPU."""
    from magr_ray_tracer_amd import dist as rdist
    Wd, Hd, frames, world = 160, 90, 2, 2
    s, sa, cam = build(lambda: scenes.sponza_class(0.2), Wd, Hd)
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    exp = np.zeros((Hd, Wd, 4), np.float32)
    total = np.zeros((Hd, Wd, 4), np.float32)
    rows = []
    for rank in range(world):
        acc = np.zeros((Hd, Wd, 4), np.float32)
        devs = []
        for p in rdist.plans("ibands", Wd, Hd, rank, world, band_rows=7):
            d = Device(Wd, Hd, y0=p["y0"], y1=p["y1"], **DEFAULT)
            d.upload(sa)
            d.set_seeds(seed_stream(p["seed_first"], p["seed_count"]))
            devs.append((d, p))
            rows += list(range(p["y0"], p["y1"]))
            o.render(cam, frames, accum=exp, seeds=seed_stream(p["seed_first"], p["seed_count"]), y0=p["y0"], y1=p["y1"])
        g = rdist.Lanes([d for d, _ in devs])
        g.render(cam, frames, each=True)
        for d, p in devs:
            a = d.read_accum()
            assert not a[:p["y0"]].any() and not a[p["y1"]:].any()      # a context writes its own band only
            acc += a
            d.close()
        total += acc
    assert sorted(rows) == list(range(Hd))
    assert_bits(total, exp, "interleaved bands vs oracle bands")


def test_contexts_share_one_device_copy_of_the_scene():
    """rt_share_scene: three contexts (two BVH2 lanes and a row band) render from the device copy the first one uploaded - each still
    bit-exact vs the oracle on its own seeds; the copy outlives the context that uploaded it; contexts that differ in accel are refused."""
    from magr_ray_tracer_amd import dist as rdist
    Wd, Hd, frames = 160, 90, 3
    s, view = scenes.sponza_class(0.2)
    sa = s.arrays()
    cam = scenes.camera_for(view, Wd, Hd)
    first = Device(Wd, Hd, **DEFAULT)
    first.upload(sa)
    lane = Device(Wd, Hd, shade_blocks_per_cu=1, persist_blocks_per_cu=4, **DEFAULT)     # another footprint, same scene
    lane.share_scene(first)
    band = Device(Wd, Hd, y0=32, y1=64, **DEFAULT)
    band.share_scene(first)
    assert lane.kernel_info()["persist"] == 1 and lane.kernel_info()["persist_grid"] < first.kernel_info()["persist_grid"]
    seeds = [seed_stream(rdist.plan("samples", Wd, Hd, 0, 1, m, 2)["seed_first"], Wd * Hd) for m in range(2)]
    first.set_seeds(seeds[0]); lane.set_seeds(seeds[1]); band.seed_default()
    for _ in range(frames):                                  # interleaved, like lanes
        for d in (first, lane, band):
            d.render(cam, 1)
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    for d, sd, name in ((first, seeds[0], "uploader"), (lane, seeds[1], "sharing lane")):
        acc, _, e, _ = o.render(cam, frames, seeds=sd.copy())     # (the oracle advances the seeds it is given)
        assert_bits(d.read_accum(), acc, name)
        assert d.counters()["extend_node_visits"] == e["node_visits"]
    accb, _, _, _ = o.render(cam, frames, y0=32, y1=64)
    assert_bits(band.read_accum()[32:64], accb[32:64], "sharing band")
    first.close()                                            # the copy stays alive with the contexts that share it
    before = lane.read_accum().copy()
    lane.render(cam, 2)
    acc, _, _, _ = o.render(cam, frames + 2, seeds=seeds[1].copy())
    assert_bits(lane.read_accum(), acc, "after the uploader was destroyed")
    assert not bits_equal(before, lane.read_accum())
    q = Device(Wd, Hd, **dict(DEFAULT, accel=1))
    with pytest.raises(RuntimeError, match="accel"):
        q.share_scene(lane)
    empty = Device(Wd, Hd, **DEFAULT)
    with pytest.raises(RuntimeError, match="no scene"):
        lane.share_scene(empty)
    for d in (lane, band, q, empty):
        d.close()


def test_bench_default_path_four_lanes_one_scene_copy_vs_oracle(tmp_path):
    """`python bench.py` as the driver runs it (four lanes, the hardware queues requested by the library, one shared device copy of the scene, HIP events on
    lane 0), at a small frame: the bench line carries what BASELINE asks for, and the accumulator it reduced is the sum of the four
    lanes' sample streams as the ORACLE renders them - bit for bit."""
    import json
    import os
    import subprocess
    import sys
    from magr_ray_tracer_amd import dist as rdist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    Wd, Hd, steps = 320, 180, 10
    dump = tmp_path / "acc.npy"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GPU_MAX_HW_QUEUES")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", str(steps), "--warmup", "0", "--no-cpu-baseline", "--no-repeat", "--no-single",
           "--width", str(Wd), "--height", str(Hd), "--detail", "0.2", "--dump-accum", str(dump)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["lanes"] == 4 and line["config"]["contexts"] == 4 and line["steps"] == steps
    assert line["metric"].startswith("Mrays/sec") and line["dtype"] == "f32" and line["vs_baseline"] is None and line["value"] > 0
    # the four lanes' streams really run side by side although nothing in this environment asked for hardware queues: the library does
    assert line["config"]["streams_concurrent"] == 4
    roof = line["roofline"]
    assert roof["job"]["lanes"] == 4 and roof["job"]["algorithmic_gbs"] > 0 and roof["job"]["ms_per_step"] > 0
    got = np.load(dump)
    s, view = scenes.sponza_class(0.2)
    sa = s.arrays(bvh4=False)
    cam = scenes.camera_for(view, Wd, Hd)
    d = Device(Wd, Hd, **DEFAULT)
    d.upload(sa)
    cam["focalLength"] = d.focus(Wd // 2, Hd // 2, cam)          # bench focuses through its first context
    d.close()
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    exp = None
    for m, frames in enumerate(rdist.lane_frames(steps, 4)):        # [3, 3, 2, 2]
        acc, _, _, _ = o.render(cam, frames, seeds=seed_stream(rdist.plan("samples", Wd, Hd, 0, 1, m, 4)["seed_first"], Wd * Hd))
        exp = acc if exp is None else exp + acc
    assert_bits(got, exp, "bench accumulator vs the oracle's four sample streams")

