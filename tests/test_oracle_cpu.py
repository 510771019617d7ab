"""CPU suite: the oracle against the reference's known answers and golden fixtures, host-side
builders, C-ABI symbol export.  No GPU needed."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

from magr_ray_tracer_amd import _lib as W, scenes
from oracle import oracle_py
from oracle.oracle_py import Oracle, seed_stream, S0, S1
from helpers import DEFAULT, assert_bits, bits_equal, build, max_rel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- known answers obtained from the reference's own functions -------------
def test_seed_stream_known_answers():
    s = seed_stream(0, 6)
    assert s.tolist() == [2274908837, 358294691, 1210119364, 2176035992, 1882851208, 698933837]
    assert np.array_equal(seed_stream(3, 3), s[3:6])


def test_rng_and_sampling_known_answers():
    L = oracle_py.lib()
    out = (C.c_float * 4)()
    s = C.c_uint32(0x12345678)
    L.orc_test_random_float3(C.byref(s), out)
    assert np.allclose(list(out), [0.52966851, 0.0834219828, 0.281752884, 0.0], rtol=0, atol=1e-9)
    s = C.c_uint32(0x12345678)
    L.orc_test_cosine_hemisphere((C.c_float * 4)(0, 1, 0, 0), C.byref(s), out)
    # one rejection round, w-lane pollution: |xyz| != 1 (Appendix B #1)
    assert np.allclose(list(out), [0.0486649163, 0.443665922, -0.357988238, -0.820144236], rtol=3e-7)
    assert s.value == 1210119364
    L.orc_test_wang_hash.restype = C.c_uint32
    assert L.orc_test_wang_hash(1) == 663891101


def test_probe_triangle_known_answer():
    L = oracle_py.lib()
    f3 = lambda *v: (C.c_float * 3)(*v)  # noqa: E731
    out = (C.c_float * 4)()
    L.orc_test_triangle(f3(0, 0, 0), f3(1, 0, 0), f3(0, 1, 0), f3(.25, .25, -1), f3(0, 0, 1), out)
    assert list(out) == [1.0, 0.25, 0.25, 7.0]


# ---- wire format ----------------------------------------------------------------------------------
def test_wire_sizes():
    exp = dict(Ray=128, ShadowRay=96, Material=80, Primitive=128, Camera=128, Settings=40, BVHNode2=48, BVHNode4=160,
               BVHInstance=68, TLASNode=48)
    for k, v in exp.items():
        assert getattr(W, k).itemsize == v
    assert W.Ray.fields["t"][1] == 96 and W.Ray.fields["pixelIdx"][1] == 108 and W.Ray.fields["u"][1] == 116
    assert W.Primitive.fields["objType"][1] == 112 and W.Material.fields["emittance"][1] == 64
    assert W.BVHNode2.fields["first"][1] == 32 and W.TLASNode.fields["leftRight"][1] == 32


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", txt)))


def test_c_abi_exports_every_declared_symbol():
    dev = W.device_lib()
    host = W.host_lib()
    d = _declared("rt355.h")
    h = _declared("rt355_host.h")
    assert len(d) >= 25 and len(h) >= 25
    for name in d:
        assert hasattr(dev, name), f"librt355.so does not export {name}"
    for name in h:
        assert hasattr(host, name), f"librt355_host.so does not export {name}"
    assert set(W.DEVICE_SYMBOLS) <= set(d) and set(W.HOST_SYMBOLS) <= set(h)


def test_device_path_fails_loudly_without_gpu():
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    from magr_ray_tracer_amd.renderer import Device, RtError
    with pytest.raises(RtError, match="no HIP device|no CPU path|hip"):
        Device(64, 36)


# ---- host builders --------------------------------------------------------------------------------
def _check_bvh2(sa, nprims_in_blas=None):
    n = sa.bvh2
    seen = np.zeros(len(sa.prims), bool)
    stack = [int(b) for b in sa.blas["bvhIdx"]]
    leaves = 0
    while stack:
        i = stack.pop()
        node = n[i]
        if node["count"] > 0:
            ids = sa.primIdx[node["first"]:node["first"] + node["count"]]
            seen[ids] = True
            leaves += 1
            for p in ids:
                pr = sa.prims[p]
                if pr["objType"] == W.PRIM_TRIANGLE:
                    v = np.stack([pr["v0"][:3], pr["v1"][:3], pr["v2"][:3]])
                    # the leaf box must overlap the triangle's box (clipped references may be tighter)
                    assert np.all(v.min(0) <= node["aabbMax"][:3] + 1e-4) and np.all(v.max(0) >= node["aabbMin"][:3] - 1e-4)
        else:
            for c in (node["first"], node["first"] + 1):
                ch = n[c]
                assert np.all(ch["aabbMin"][:3] >= node["aabbMin"][:3] - 1e-5) and np.all(ch["aabbMax"][:3] <= node["aabbMax"][:3] + 1e-5)
                stack.append(int(c))
        assert node["aabbMin"][3] == 0 and node["aabbMax"][3] == 0
    return seen, leaves


def test_bvh2_sah_structure():
    s, view = scenes.bunny_class(16)
    sa = s.arrays()
    seen, leaves = _check_bvh2(sa)
    assert seen.all()
    st = s.stats()
    assert st["spatial_splits"] == 0 and len(sa.primIdx) == len(sa.prims)       # alpha = 1: plain SAH, a permutation
    assert sorted(sa.primIdx.tolist()) == list(range(len(sa.prims)))
    assert len(sa.bvh2) == st["nodes"] and len(sa.bvh2) % 2 == 1
    # LIFO build order: the right child's subtree is emitted first, so the first leaf in primIdx order sits under a right child
    root = sa.bvh2[0]
    assert root["count"] == 0 and root["first"] == 1


def test_sbvh_spatial_splits_duplicate_references():
    s, view = scenes.bunny_class(12, alpha=0.0)
    sa = s.arrays()
    seen, _ = _check_bvh2(sa)
    st = s.stats()
    assert seen.all() and st["spatial_splits"] > 0 and st["prims_clipped"] > 0
    assert len(sa.primIdx) > len(sa.prims)          # clipped triangles are referenced from both sides


def test_bvh_variants_agree_on_hits():
    """BVH2 (alpha=1), SBVH (alpha=0) and the BVH4 collapse return identical nearest hits."""
    W_, H_ = 96, 54
    res = []
    for alpha in (1.0, 0.0):
        s, view = scenes.bunny_class(14, alpha=alpha)
        sa = s.arrays()
        cam = scenes.camera_for(view, W_, H_)
        for accel in (0, 1):
            o = Oracle(sa, W_, H_, **dict(DEFAULT, accel=accel))
            seeds = seed_stream(0, W_ * H_)
            rays = o.generate(cam, 0, W_ * H_, seeds)
            steps, ctr = o.extend(rays, want_steps=True)
            res.append((rays["t"].copy(), rays["primIdx"].copy(), ctr))
    for t, p, _ in res[1:]:
        assert np.array_equal(p, res[0][1])
        assert bits_equal(t, res[0][0])
    assert res[1][2]["node_visits"] < res[0][2]["node_visits"]   # 4-wide: fewer node fetches


def test_bvh4_collapse_layout():
    s, view = scenes.bunny_class(10)
    sa = s.arrays()
    assert len(sa.bvh4) == len(sa.bvh2)             # sparse: same index space as the BVH2 array
    leaf_slots = sa.bvh2["count"] > 0
    assert np.all(sa.bvh4["first"][leaf_slots] == 0) and np.all(sa.bvh4["count"][leaf_slots] == 0)
    # every primitive reachable exactly once from the root
    stack, cnt = [0], 0
    while stack:
        n = sa.bvh4[stack.pop()]
        for k in range(4):
            if n["first"][k] == -1:
                assert n["count"][k] == -1
                continue
            if n["count"][k] > 0:
                cnt += int(n["count"][k])
            else:
                stack.append(int(n["first"][k]))
    assert cnt == len(sa.primIdx)


def test_tlas_two_blas():
    s, view = scenes.two_blas_scene(alpha=1.0, n=10)
    sa = s.arrays()
    assert len(sa.blas) == 2 and len(sa.tlas) == 4
    root = sa.tlas[0]
    assert root["leftRight"] == (1 | (2 << 16))
    assert sa.tlas[1]["leftRight"] == 0 and sa.tlas[2]["leftRight"] == 0 and sa.tlas[2]["BLASidx"] == 1
    assert np.allclose(sa.blas["invT"][0].reshape(4, 4), np.eye(4))
    s1, _ = scenes.cube_scene()
    t1 = s1.arrays().tlas
    assert len(t1) == 2 and t1[0]["leftRight"] == 0   # single BLAS: root is a leaf


def test_scene_factory_matches_reference_formulas():
    s, _ = scenes.cube_scene()
    sa = s.arrays()
    tri = sa.prims[sa.prims["objType"] == W.PRIM_TRIANGLE]
    v0, v1, v2 = tri["v0"][:, :3], tri["v1"][:, :3], tri["v2"][:, :3]
    n = np.cross(v1 - v0, v2 - v0)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    assert np.allclose(tri["N"][:, :3], n, atol=1e-6) and np.all(tri["N"][:, 3] == 0)
    assert np.allclose(tri["centroid"][:, :3], (v0 + v1 + v2) / 3, atol=1e-6)
    area = 0.5 * np.linalg.norm(np.cross(v1 - v0, v2 - v0), axis=1)
    assert np.allclose(tri["area"], area, rtol=1e-4)
    lights = sa.lights
    assert len(lights) == 2 and np.all(sa.mats[sa.prims[lights]["matIdx"]]["isLight"] == 1)


def test_camera_basis():
    cam = scenes.make_camera(1280, 720, (-10, 10, 15), (0, 0, 1), fov=110.0)
    assert np.allclose(cam["right"][:3], [1, 0, 0]) and np.allclose(cam["up"][:3], [0, -1, 0])   # up = cross(right, fwd)
    vh = 2 * np.tan(np.deg2rad(110.0) / 2)
    assert np.isclose(np.linalg.norm(cam["vertical"][:3]), vh, rtol=1e-6)
    assert np.isclose(np.linalg.norm(cam["horizontal"][:3]), vh * 1280 / 720, rtol=1e-6)
    tl = cam["origin"] - cam["horizontal"] / 2 - cam["vertical"] / 2 - cam["forward"]
    assert np.allclose(cam["topLeft"], tl, atol=1e-5)


# ---- oracle behaviour ---------------------------------------------------------------------------------
def test_oracle_is_deterministic_and_state_carries_over():
    s, sa, cam = build(scenes.cube_scene, 64, 36)
    o = Oracle(sa, 64, 36, **DEFAULT)
    a2, s2, *_ = o.render(cam, 2)
    a1, s1, *_ = o.render(cam, 1)
    a1b, s1b, *_ = o.render(cam, 1, accum=a1, seeds=s1)      # accumulate a second frame on top
    assert bits_equal(a2, a1b) and np.array_equal(s2, s1b)
    assert a2[..., :3].mean() > 0.05


def test_oracle_schedules_differ_but_agree_statistically():
    s, sa, cam = build(scenes.cube_scene, 48, 27)
    a1, *_ = Oracle(sa, 48, 27, **DEFAULT, schedule=S1).render(cam, 24)
    a0, *_ = Oracle(sa, 48, 27, **DEFAULT, schedule=S0).render(cam, 24)
    assert not bits_equal(a1, a0)
    m1, m0 = a1[..., :3].mean(), a0[..., :3].mean()
    assert abs(m1 - m0) / m1 < 0.05


def test_oracle_band_equals_itself_and_covers_rows_only():
    s, sa, cam = build(scenes.cube_scene, 64, 36)
    o = Oracle(sa, 64, 36, **DEFAULT)
    a, *_ = o.render(cam, 2, y0=10, y1=20)
    assert np.all(a[:10] == 0) and np.all(a[20:] == 0) and a[10:20, :, :3].sum() > 0


def test_rr_off_connects_per_bounce_same_sum_order_independent_of_queue():
    """Without Russian roulette the host connects after every bounce (renderer.cpp:85-87)."""
    s, sa, cam = build(scenes.cube_scene, 48, 27)
    a, *_ = Oracle(sa, 48, 27, **dict(DEFAULT, russian_roulette=False)).render(cam, 2)
    b, *_ = Oracle(sa, 48, 27, **DEFAULT).render(cam, 2)
    assert a[..., :3].mean() > 0 and not bits_equal(a, b)


# ---- golden fixtures produced by the REFERENCE's own kernels on the MI355X ---------------------------------
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_*.npz")))


class _SA:
    pass


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_reference_kernels(path):
    """Inputs and outputs were captured by tests/golden/make_golden.py running the reference's OpenCL
    kernels (oracle/_ref) on the GPU.  Integer/index results and everything extend() computes must be
    bit-exact; quantities behind the library normalize()/length() may differ in the last ulps."""
    g = np.load(path)
    sa = _SA()
    for k in ("prims", "mats", "tex", "lights", "bvh2", "bvh4", "primIdx", "tlas", "blas"):
        setattr(sa, k, g[k])
    v = {k: int(g["variant"][i]) for i, k in enumerate(("shading", "sampling", "accel", "russian_roulette", "filter_fireflies"))}
    Wd, Hd, n, rows = (int(x) for x in g["dims"])
    o = Oracle(sa, Wd, Hd, **v, schedule=S0)
    cam = g["cam"].view(W.Camera)[0]
    seeds = g["seeds_in"].copy()
    rays = o.generate(cam, 0, n, seeds)
    ref = g["gen_rays"].view(W.Ray)
    assert_bits(rays["O"], ref["O"], "generate O")
    assert np.array_equal(seeds, g["gen_seeds"])
    assert np.abs(rays["D"] - ref["D"]).max() < 1e-6      # unit-scale direction: a few ulp (hardware rsqrt)
    # extend on the reference's own generated rays: bit-exact
    rays = ref.copy()
    o.extend(rays)
    ext = g["ext_rays"].view(W.Ray)
    hit = ext["primIdx"] != -1
    for f in ("t", "primIdx", "I", "N"):
        assert_bits(rays[f], ext[f], "extend " + f)
    assert_bits(rays["u"][hit], ext["u"][hit], "extend u")
    assert_bits(rays["v"][hit], ext["v"][hit], "extend v")
    # shade, schedule S0 (the reference launched with one work-item)
    accum = np.zeros((Hd * Wd, 4), np.float32)
    seeds = g["gen_seeds"].copy()
    out, sh = o.shade(ext.copy(), accum, seeds)
    rout, rsh = g["shade_rays"].view(W.Ray), g["shade_shadow"].view(W.ShadowRay)
    assert len(out) == len(rout) and len(sh) == len(rsh)
    assert seeds[0] == g["shade_seeds"][0]
    for f in ("pixelIdx", "bounces", "inside", "lastSpecular"):
        assert np.array_equal(out[f], rout[f]), f
    assert max_rel(out["D"], rout["D"], 1e-2) < 1e-4 and max_rel(out["O"], rout["O"], 1e-2) < 1e-4
    assert max_rel(out["intensity"], rout["intensity"], 1e-3) < 1e-5
    if len(sh):
        for f in ("lightIdx", "pixelIdx"):
            assert np.array_equal(sh[f], rsh[f]), f
        assert_bits(sh["I"], rsh["I"], "shadow I")
        assert_bits(sh["BRDF"], rsh["BRDF"], "shadow BRDF")
        assert max_rel(sh["L"], rsh["L"], 1e-2) < 1e-4 and max_rel(sh["dist"], rsh["dist"]) < 1e-6
    assert max_rel(accum[:rows * Wd], g["shade_accum"].reshape(-1, 4), 1e-4) < 1e-5 and not accum[rows * Wd:].any()
    # connect (S0) on the reference's own shadow rays
    if len(rsh):
        accum = np.zeros((Hd * Wd, 4), np.float32)
        o.connect(rsh.copy(), accum)
        assert max_rel(accum[:rows * Wd], g["connect_accum"].reshape(-1, 4), 1e-4) < 1e-5


# ---- BASELINE config 1: cube, 256x256, 1 spp, CPU template renderer + SAH build (plumbing, no GPU) --------------------
def test_config1_cube_256_cpu_plumbing():
    import time
    Wd = Hd = 256
    s, view = scenes.cube_scene()
    sa = s.arrays()
    st = s.stats()
    assert st["prims"] == 16 and st["nodes"] == len(sa.bvh2) and st["depth"] >= 3
    cam = scenes.camera_for(view, Wd, Hd)
    o = Oracle(sa, Wd, Hd, **DEFAULT)
    # CPU-B: the upstream template's trace loop shape (one primary ray per pixel, nearest hit, normal visualisation)
    t0 = time.perf_counter()
    img = o.trace_normals(cam, threads=2)
    dt = time.perf_counter() - t0
    hit = img[..., :3].any(axis=2)
    assert 0.3 < hit.mean() <= 1.0
    n = img[hit][:, :3] * 2 - 1
    assert np.allclose(np.linalg.norm(n, axis=1), 1, atol=1e-5)          # unit normals of axis-aligned faces
    assert set(np.unique(np.round(n, 3))) <= {-1.0, 0.0, 1.0}
    mrays = Wd * Hd / dt / 1e6                                           # the reference's "Mrays/s" formula (renderer.cpp:60-62)
    assert mrays > 0.01
    # CPU-A: one sample per pixel of the full wavefront path
    acc, seeds, e, c = o.render(cam, 1)
    assert e["rays"] >= Wd * Hd and c["rays"] > 0 and acc[..., :3].mean() > 0.01
    # same primary visibility from both paths
    rays = o.generate(cam, 0, Wd * Hd, seed_stream(0, Wd * Hd), antiAliasing=0)
    o.extend(rays)
    # (generate() jitters the lens even without AA, so compare statistically)
    assert abs((rays["primIdx"] != -1).mean() - hit.mean()) < 0.01


@pytest.mark.parametrize("alpha", [1.0, 1e-5, 0.0])
def test_parallel_build_is_index_exact(alpha):
    """the task-parallel builder numbers nodes in the reference's LIFO order -> identical arrays."""
    from magr_ray_tracer_amd.scene import Scene

    def build(threads):
        s = Scene()
        scenes._std_materials(s)

        def blob(U, V):
            th, ph = U * 2 * np.pi, V * np.pi
            r = 1.0 + 0.18 * np.sin(5 * th) * np.sin(3 * ph) ** 2
            return r * np.sin(ph) * np.cos(th), 1.25 + r * np.cos(ph), r * np.sin(ph) * np.sin(th)
        s.AddTriangles(scenes.param_surface(blob, 56, 56), "sand")
        s.AddQuad((-10, 0, -10), (-10, 0, 10), (10, 0, 10), (10, 0, -10), "grey")
        s.BuildBLAS(0, alpha, threads=threads)
        return s.arrays(bvh4=False), s.stats()
    a1, st1 = build(1)
    a4, st4 = build(4)
    assert a1.bvh2.tobytes() == a4.bvh2.tobytes() and np.array_equal(a1.primIdx, a4.primIdx)
    for k in ("nodes", "depth", "spatial_splits", "prims_clipped"):
        assert st1[k] == st4[k], k


# ---- whole frames of the reference's own kernels (renderer.cpp:64-94 launch sequence, schedule S0), captured on the MI355X -------
FRAMES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "refframe_*.npz")))


@pytest.mark.parametrize("path", FRAMES, ids=[os.path.basename(p) for p in FRAMES])
def test_oracle_follows_reference_frames_launch_by_launch(path):
    """Every launch of a whole reference frame (7 x extend/shade, connect per bounce or deferred) replayed through the oracle from
    the reference's rays and RNG state: see helpers.teacher_forced_s0.  The fixtures hold a band that drives every shading
    branch; the branch counts recomputed from the oracle's rays must equal the ones recorded from the reference's."""
    from helpers import branch_counts, load_frame_fixture, teacher_forced_s0
    sa, v, cam, (Wd, Hd, y0, y1), cap, heat, bc = load_frame_fixture(path)
    o = Oracle(sa, Wd, Hd, **v, schedule=S0)
    mine = []
    stats = teacher_forced_s0(o, cap, sa, os.path.basename(path), full_rays=False, collect=mine)
    got = branch_counts(dict(ext=mine, shadow=cap["shadow"], last_out=cap["last_out"]), sa)
    assert got == bc, (got, bc)
    if "branch" in os.path.basename(path) and "_free_" not in os.path.basename(path):
        need = ["light_spec", "tex_tri", "tex_sphere", "inside", "inside_dielectric", "tir", "last_bounce"]
        if v["shading"] == 1:     # Kajiya has no shadow rays and never sets lastSpecular on a child ray (shading.cl:7-70)
            need += ["light_spec_later", "light_nospec", "sphere_light_shadow", "tri_light_shadow"]
        assert all(bc[k] > 0 for k in need), bc
    if "_free_" in os.path.basename(path):
        # a band on which no knife-edge decision flips: the oracle, running freely from the seeds, stays on the reference's frame
        from helpers import compare_frames_s0, oracle_frame_s0
        print(compare_frames_s0(cap, oracle_frame_s0(o, cam, y0, y1), os.path.basename(path)))
    # the reference's own heat-map values of the primary rays: accum[slot] = steps / 255.f (wavefront.cl:66-67)
    n = (y1 - y0) * Wd
    seeds = seed_stream(y0 * Wd, n)
    rays = o.generate(cam, y0 * Wd, n, seeds)
    steps, _ = o.extend(rays, want_steps=True)
    assert_bits(steps.astype(np.float32) / np.float32(255.0), heat, "steps / 255 of the primary rays")
    assert stats["rays"] > 3000


def test_oracle_postproc_matches_reference_postproc_kernels():
    """tests/golden/refpost.npz: outputs of the reference's own postproc.cl kernels (compiled for gfx950, run on the MI355X by
    tests/golden/make_golden.py) for the parameter sets of helpers.POST_SETS.  prep and chromatic are + - * / fma: bit for bit;
    vignetting goes through length() = the hardware v_sqrt_f32 and gammaCorr through the device library's pow(): a few ulp."""
    from helpers import POST_SETS, REF_H, REF_W, post_test_accum
    path = os.path.join(ROOT, "tests", "golden", "refpost.npz")
    if not os.path.exists(path):
        pytest.skip("refpost.npz not generated yet")
    g = np.load(path)
    rows = int(g["rows"])
    assert np.array_equal(g["params"], np.array(POST_SETS, np.float64))
    full = np.zeros((REF_H, REF_W, 4), np.float32)
    full.reshape(-1, 4)[:rows * REF_W] = post_test_accum(rows)
    for k, (frames, vignette, gamma, chromatic) in enumerate(POST_SETS):
        f, b8 = oracle_py.postproc(full, frames, vignette, gamma, chromatic)
        o = f.reshape(-1, 4)[:rows * REF_W, :3]
        ref = np.minimum(g[f"out{k}"], np.float32(1.0))
        if gamma == 1.0 and vignette == 0.0:
            assert_bits(o, ref, f"postproc set {k}")
        else:   # pow() and the hardware sqrt inside length() (vignetting) have no bit-exact CPU counterpart
            assert max_rel(o, ref, 1e-6) < 3e-6, k


def _instanced_pair(moved):
    """Two boxes on a floor.  moved=True: box 2 is modelled around the origin and placed by its instance transform (invT = world ->
    instance); moved=False: the same box with its vertices pre-transformed into world space, identity instance."""
    from magr_ray_tracer_amd.scenes import Scene, _std_materials, box_tris
    a = np.deg2rad(31.0)
    T = np.array([[np.cos(a), 0, -np.sin(a), 1.6], [0, 1, 0, 0.25], [np.sin(a), 0, np.cos(a), -0.4], [0, 0, 0, 1]], np.float64)   # instance -> world
    s = Scene()
    _std_materials(s)
    s.AddTriangles(box_tris((-2.4, 0, -0.8), (-0.9, 1.4, 0.6)), "red")
    s.AddQuad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), "grey")
    s.AddQuad((-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1), "white-light")
    s.BuildBLAS(0, 1.0)
    st = s.num_prims
    tris = box_tris((-0.7, 0.0, -0.5), (0.7, 1.6, 0.5)).astype(np.float64)
    if not moved:
        tris = (tris.reshape(-1, 3) @ T[:3, :3].T + T[:3, 3]).reshape(-1, 3, 3)
    s.AddTriangles(tris.astype(np.float32), "green")
    s.BuildBLAS(st, 1.0)
    if moved:
        s.SetInstanceTransform(1, np.linalg.inv(T).astype(np.float32))
    return s.arrays(), T


def test_tlas_leaf_bounds_follow_the_instance_transform():
    """The reference copies the BLAS root's object-space box into the TLAS leaf (tlas.cpp:15-17): right for its identity instances
    only.  A moved instance must get the world-space bounds of its box, or the world ray is slab-tested against the wrong box and
    the instance is culled: the moved instance has to render like the same geometry pre-transformed into world space."""
    sa_m, T = _instanced_pair(True)
    sa_w, _ = _instanced_pair(False)
    # identity instance 0: the reference's copy, bit for bit
    leaves = {int(n["BLASidx"]): n for n in sa_m.tlas[1:] if n["leftRight"] == 0}
    root0 = sa_m.bvh2[sa_m.blas["bvhIdx"][0]]
    assert_bits(leaves[0]["aabbMin"], root0["aabbMin"], "identity leaf min")
    assert_bits(leaves[0]["aabbMax"], root0["aabbMax"], "identity leaf max")
    # moved instance 1: the leaf contains every world-space vertex and is not much larger than their bounds
    world = sa_w.prims["v0"][-12:, :3].tolist() + sa_w.prims["v1"][-12:, :3].tolist() + sa_w.prims["v2"][-12:, :3].tolist()
    world = np.array(world)
    lo, hi = leaves[1]["aabbMin"][:3], leaves[1]["aabbMax"][:3]
    assert (world >= lo - 1e-6).all() and (world <= hi + 1e-6).all()
    assert np.all(lo > world.min(0) - 1e-3) and np.all(hi < world.max(0) + 1e-3)
    # and the two scenes look the same to the primary rays
    Wd, Hd = 160, 90
    view = dict(origin=(0.4, 2.2, 5.5), forward=(0.05, 0.2, 0.97), fov=65.0, aperture=0.0)
    cam = scenes.camera_for(view, Wd, Hd)
    hits = []
    for sa in (sa_m, sa_w):
        o = Oracle(sa, Wd, Hd, **DEFAULT)
        seeds = seed_stream(0, Wd * Hd)
        rays = o.generate(cam, 0, Wd * Hd, seeds)
        o.extend(rays)
        hits.append(rays)
    a, b = hits
    on_box = (b["primIdx"] >= len(sa_w.prims) - 12)
    assert on_box.sum() > 300                                  # the moved box is in view ...
    same = a["primIdx"] == b["primIdx"]
    assert same.mean() > 0.998 and (a["primIdx"][on_box] == b["primIdx"][on_box]).mean() > 0.98   # ... and is hit (edges may flip)
    assert max_rel(a["t"][same & on_box], b["t"][same & on_box], 1e-3) < 1e-4


def test_tlas_refuses_more_than_256_instances():
    from magr_ray_tracer_amd.scenes import Scene, _std_materials
    s = Scene()
    _std_materials(s)
    for k in range(257):
        st = s.num_prims
        s.AddTriangle((k, 0, 0), (k + 0.5, 0, 0), (k, 0.5, 0), "red")
        s.BuildBLAS(st, 1.0)
    with pytest.raises(RuntimeError, match="256"):
        s.arrays(bvh4=False)


def test_upload_validation_without_a_gpu_refuses_oversized_and_malformed_tlas():
    """rt_validate_scene = the host-side checks rt_upload_scene runs before it touches the device.  TLAS child ids and instance ids
    travel as 15-bit values on the traversal stacks (bit 15 = leaf), so a caller-provided TLAS with more than 32768 nodes or instances
    would pop the wrong node: refused with RT_E_UNSUPPORTED.  Also: the same call accepts a scene the repo's own builders made, and
    refuses a TLAS cycle and an out-of-range child."""
    lib = W.device_lib()
    s, _ = scenes.two_blas_scene(0.0, 8)
    sa = s.arrays()

    def validate(tlas, blas, accel=0):
        nodes = sa.nodes(accel)
        P = W.ptr
        return lib.rt_validate_scene(accel, P(sa.prims), len(sa.prims), P(sa.mats), len(sa.mats), None, 0, P(sa.lights), len(sa.lights),
                                     P(nodes), len(nodes), P(sa.primIdx), len(sa.primIdx), P(tlas), len(tlas), P(blas), len(blas))
    assert validate(sa.tlas, sa.blas) == 0 and validate(sa.tlas, sa.blas, 1) == 0
    # 32769 instances of BLAS 0 under a right-leaning chain of TLAS nodes would need 16-bit ids: refused before anything else is looked at
    big_blas = np.repeat(sa.blas[:1], 0x8001)
    rc = validate(sa.tlas, big_blas)
    assert rc == -4 and b"32768" in lib.rt_last_error()
    big_tlas = np.zeros(0x8001, dtype=sa.tlas.dtype)
    rc = validate(big_tlas, sa.blas)
    assert rc == -4 and b"32768" in lib.rt_last_error()
    cyc = sa.tlas.copy()
    inner = 0                                                   # node 0 is the root the kernels start from (tlas.cpp:38)
    cyc["leftRight"][inner] = (inner << 16) | (int(cyc["leftRight"][inner]) & 0xffff)      # right child = itself
    assert validate(cyc, sa.blas) == -1 and b"twice" in lib.rt_last_error()
    bad = sa.tlas.copy()
    bad["leftRight"][inner] = (len(bad) << 16) | 1
    assert validate(bad, sa.blas) == -1 and b"out of range" in lib.rt_last_error()


def test_bvh4_collapse_of_the_reference_13_node_fixture():
    """The one builder fixture the reference holds: the hand-built 13-node BVH2 of src/bvh.cpp:615-674 (disabled `#if 0` debugging
    input of BVH4::BVH4, no expected output recorded).  The collapse BVH4::Convert / Collapse (bvh.cpp:695-787) prescribes for it,
    worked out by hand:

        BVH2:  0 -> (1, 2)   1 -> (3, leaf4)   2 -> (leaf5, 6)   3 -> (leaf7, leaf8)   6 -> (9, leaf10)   9 -> (leaf11, leaf12)
               boxes +-20, +-9, +-12, +-8, +-10, +-9 for nodes 0, 1, 2, 3, 6, 9; leaf k has count = first = k (leaf 12: first 0)
        Convert: every interior node gets its two children in slots 0, 1 (leaf: its first/count, interior: node id / 0).
        Collapse(0): slots (1, 2), both adoptable; half areas 3*18^2 = 972 and 3*24^2 = 1728 -> adopt node 2: (1, leaf5, 6);
                     candidates 1 (972) and 6 (3*20^2 = 1200) -> adopt node 6: (1, leaf5, 9, leaf10); full -> stop.
        Collapse(1): (3, leaf4) -> adopt node 3: (leaf7, leaf4, leaf8).   Collapse(9): two leaves, nothing to adopt.
        Nodes 2, 3 and 6 are absorbed but stay in the array as Convert left them (the array keeps the BVH2's index space).

    This pins the restatement's collapse rule (greedy largest-area adoption, slot placement, recursion order); the builders'
    node-array parity with the reference's own bvh.cpp output stays UNPINNED - building that file needs stand-in headers."""
    n2 = np.zeros(13, dtype=W.BVHNode2)

    def interior(i, first, half):
        n2["aabbMin"][i][:3], n2["aabbMax"][i][:3] = -half, half
        n2["first"][i], n2["count"][i] = first, 0
    interior(0, 1, 20), interior(1, 3, 9), interior(2, 5, 12), interior(3, 7, 8), interior(6, 9, 10), interior(9, 11, 9)
    for k in (4, 5, 7, 8, 10, 11):
        n2["first"][k], n2["count"][k] = k, k
    n2["first"][12], n2["count"][12] = 0, 12           # the fixture sets `count` twice and never `first`
    out = np.zeros(13, dtype=W.BVHNode4)
    assert W.host_lib().rth_bvh4_from_nodes(W.ptr(n2), 13, W.ptr(out)) == 0
    INV = -1
    assert out["first"][0].tolist() == [1, 5, 9, 10] and out["count"][0].tolist() == [0, 5, 0, 10]
    assert out["first"][1].tolist() == [7, 4, 8, INV] and out["count"][1].tolist() == [7, 4, 8, INV]
    assert out["first"][9].tolist() == [11, 0, INV, INV] and out["count"][9].tolist() == [11, 12, INV, INV]
    # child boxes travel with the adopted children
    assert out["aabbMax"][0][0][:3].tolist() == [9, 9, 9] and out["aabbMax"][0][2][:3].tolist() == [9, 9, 9]
    assert out["aabbMax"][0][1][:3].tolist() == [0, 0, 0] and out["aabbMin"][0][2][:3].tolist() == [-9, -9, -9]
    # absorbed nodes keep their Convert() state; leaf slots of the array stay zero-filled
    assert out["first"][2].tolist() == [5, 6, INV, INV] and out["count"][2].tolist() == [5, 0, INV, INV]
    assert out["first"][3].tolist() == [7, 8, INV, INV] and out["first"][6].tolist() == [9, 10, INV, INV] and out["count"][6].tolist() == [0, 10, INV, INV]
    for k in (4, 5, 7, 8, 10, 11, 12):
        assert not out[k].tobytes().strip(b"\0")
