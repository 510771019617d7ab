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
