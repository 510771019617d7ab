"""Generates tests/golden/ref_*.npz ON THE GPU BOX by running the REFERENCE's own OpenCL kernels
(oracle/_ref/*.co, compiled from /root/reference by oracle/build_ref.sh; nothing of the reference's
source is stored here).  Each fixture holds the inputs (scene arrays, camera, seeds) and the reference's
outputs per stage:  generate -> extend -> shade (one work-item = schedule S0) -> connect (S0); the refframe_*.npz fixtures hold
WHOLE frames: the reference's generate -> 7 x (extend, shade[, connect]) -> [connect] launch sequence (src/renderer.cpp:64-94) over a
thin band of the frame that drives every shading branch, with the per-bounce queues and the final accumulator.

    gpurun -- python tests/golden/make_golden.py gpurun_out/golden     # then copy the .npz into tests/golden/
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from magr_ray_tracer_amd import scenes  # noqa: E402
from oracle.oracle_py import seed_stream  # noqa: E402
from ref_gpu import REF_H, REF_W, RefGPU  # noqa: E402

CASES = {
    # name: (scene factory, view override, rows, variant)
    "mixed_nee": (lambda: scenes.mixed_scene(textured=True), dict(forward=(0.32, 0.75, 0.92)), 4, dict()),
    "mixed_kajiya_hemi": (lambda: scenes.mixed_scene(textured=True), dict(forward=(0.32, 0.75, 0.92)), 4,
                          dict(shading=0, sampling=0, russian_roulette=False, filter_fireflies=False)),
    "mixed_nee_hemi_norr": (lambda: scenes.mixed_scene(textured=True), dict(forward=(0.32, 0.75, 0.92)), 4,
                            dict(sampling=0, russian_roulette=False)),
    "mixed_kajiya_cos_rr": (lambda: scenes.mixed_scene(textured=True), dict(forward=(0.32, 0.75, 0.92)), 4, dict(shading=0)),
    "cube_nee_bvh4": (scenes.cube_scene, dict(forward=(0.5, 0.9, 0.82)), 4, dict(accel=1)),
    "twoblas_nee": (lambda: scenes.two_blas_scene(alpha=0.0, n=10), dict(forward=(0.02, 0.8, 0.97)), 4, dict()),
}
VKEYS = ("shading", "sampling", "accel", "russian_roulette", "filter_fireflies")
VDEF = dict(shading=1, sampling=1, accel=0, russian_roulette=True, filter_fireflies=True)


def make(name, outdir):
    fn, vo, rows, variant = CASES[name]
    variant = dict(VDEF, **variant)
    s, view = fn()
    view = dict(view, **vo)
    sa = s.arrays()
    cam = scenes.camera_for(view, REF_W, REF_H)
    ref = RefGPU(sa, **variant)
    cam["focalLength"] = ref.focus(REF_W // 2, rows // 2, cam)
    n = REF_W * rows
    seeds_in = seed_stream(0, n)
    gen_rays, gen_seeds = ref.generate(cam, seeds_in)
    ext_rays = ref.extend(gen_rays)
    ref.clear_accum()
    shade_rays, shade_shadow, shade_seeds = ref.shade_s0(ext_rays, gen_seeds)
    shade_accum = ref.read_accum(rows)
    ref.clear_accum()
    if len(shade_shadow):
        ref.connect_s0(shade_shadow)
    connect_accum = ref.read_accum(rows)
    ref.close()
    path = os.path.join(outdir, f"ref_{name}.npz")
    np.savez_compressed(
        path, prims=sa.prims, mats=sa.mats, tex=sa.tex, lights=sa.lights, bvh2=sa.bvh2, bvh4=sa.bvh4, primIdx=sa.primIdx,
        tlas=sa.tlas, blas=sa.blas, cam=np.ascontiguousarray(cam).reshape(1).view(np.uint8), seeds_in=seeds_in,
        variant=np.array([int(variant[k]) for k in VKEYS], np.int32), dims=np.array([REF_W, REF_H, n, rows], np.int32),
        gen_rays=gen_rays.view(np.uint8), gen_seeds=gen_seeds, ext_rays=ext_rays.view(np.uint8),
        shade_rays=shade_rays.view(np.uint8), shade_shadow=shade_shadow.view(np.uint8), shade_seeds=shade_seeds[:4],
        shade_accum=shade_accum, connect_accum=connect_accum)
    print(name, "rays", n, "hits", int((ext_rays["primIdx"] != -1).sum()), "ext", len(shade_rays), "shadow", len(shade_shadow),
          "->", path, os.path.getsize(path) // 1024, "KiB", flush=True)


# ---- whole frames: the reference's generate -> 7 x (extend, shade[, connect]) -> [connect] sequence (renderer.cpp:64-94), S0 ----
FRAME_CASES = {
    # name: (scene factory, view override, (y0, y1), variant)
    "branch_nee": (scenes.branch_scene, dict(), (359, 361), dict()),
    "branch_nee_hemi_norr": (scenes.branch_scene, dict(), (359, 361), dict(sampling=0, russian_roulette=False, filter_fireflies=False)),
    "branch_kajiya": (scenes.branch_scene, dict(), (359, 361), dict(shading=0)),
    "branch_nee_bvh4": (scenes.branch_scene, dict(), (359, 361), dict(accel=1)),
    "branch_fisheye": (scenes.branch_scene, dict(type=1, fov=75.0), (359, 361), dict()),
    "twoblas": (lambda: scenes.two_blas_scene(alpha=0.0, n=10), dict(), (367, 369), dict()),
    # one-row bands on which the free-running comparison meets no knife-edge decision (tools/find_flipfree.py)
    "branch_free_nee": (scenes.branch_scene, dict(), (359, 360), dict()),
    "branch_free_nee_hemi_norr": (scenes.branch_scene, dict(), (360, 361), dict(sampling=0, russian_roulette=False, filter_fireflies=False)),
    "branch_free_kajiya": (scenes.branch_scene, dict(), (360, 361), dict(shading=0)),
}


def _flags(r):
    return (r["bounces"].astype(np.uint8) | ((r["inside"] != 0).astype(np.uint8) << 4) | ((r["lastSpecular"] != 0).astype(np.uint8) << 5))


def make_frame(name, outdir):
    """Fixture = inputs + what the reference's kernels made of them at every launch, reduced to the fields the comparison reads
    (69 B per ray instead of 128): O, D, intensity entering each extend; t, primIdx, u, v leaving it; pixel and flag words."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import branch_counts
    fn, vo, (y0, y1), variant = FRAME_CASES[name]
    variant = dict(VDEF, **variant)
    s, view = fn()
    view = dict(view, **vo)
    sa = s.arrays()
    cam = scenes.camera_for(view, REF_W, REF_H)
    ref = RefGPU(sa, **variant)
    cam["focalLength"] = ref.focus(REF_W // 2, (y0 + y1) // 2, cam)
    cap = ref.frame_s0(cam, y0, y1, shading=variant["shading"], russian_roulette=variant["russian_roulette"])
    # the reference's own heat-map values of the band's primary rays (renderBVH, wavefront.cl:66-67): accum[slot] = steps / 255.f
    _, heat = ref.extend(cap["gen"], renderBVH=True)
    ref.close()
    bc = branch_counts(cap, sa)
    sh = cap["shadow"] or []
    shcat = np.concatenate(sh) if len(sh) else np.zeros(0, dtype=np.dtype([("lightIdx", "<i4"), ("pixelIdx", "<i4")]))
    cat = np.concatenate(cap["ext"] + [cap["last_out"]])
    path = os.path.join(outdir, f"refframe_{name}.npz")
    np.savez_compressed(
        path, prims=sa.prims, mats=sa.mats, tex=sa.tex, lights=sa.lights, bvh2=sa.bvh2, bvh4=sa.bvh4, primIdx=sa.primIdx,
        tlas=sa.tlas, blas=sa.blas, cam=np.ascontiguousarray(cam).reshape(1).view(np.uint8),
        variant=np.array([int(variant[k]) for k in VKEYS], np.int32), dims=np.array([REF_W, REF_H, y0, y1], np.int32),
        gen_seeds=cap["gen_seeds"], n_in=np.array(cap["n_in"], np.int32), n_out=np.array(cap["n_out"], np.int32),
        n_shadow=np.array(cap["n_shadow"], np.int32), seed0=np.array(cap["seed0"], np.uint32),
        shadow_len=np.array([len(x) for x in sh], np.int32),
        shadow_lightIdx=shcat["lightIdx"].astype(np.int32), shadow_pixelIdx=shcat["pixelIdx"].astype(np.int32),
        ray_O=np.ascontiguousarray(cat["O"]), ray_D=np.ascontiguousarray(cat["D"]), ray_intensity=np.ascontiguousarray(cat["intensity"]),
        ray_t=cat["t"].astype(np.float32), ray_primIdx=cat["primIdx"].astype(np.int32), ray_u=cat["u"].astype(np.float32),
        ray_v=cat["v"].astype(np.float32), ray_pixelIdx=cat["pixelIdx"].astype(np.int32), ray_flags=_flags(cat),
        accum=cap["accum"], heat=np.ascontiguousarray(heat[:, 0]),
        branch_names=np.array(sorted(bc)), branch_counts=np.array([bc[k] for k in sorted(bc)], np.int64))
    print("frame", name, "rows", (y0, y1), "rays/bounce", cap["n_in"], "shadow", cap["n_shadow"][-1], bc, "->", path,
          os.path.getsize(path) // 1024, "KiB", flush=True)


def make_post(outdir):
    """refpost.npz: the reference's postproc.cl kernels (prep -> [vignetting] -> [gammaCorr] -> [chromatic], renderer.cpp:95-124)
    on the deterministic accumulator band of helpers.post_test_accum, one output per parameter set of helpers.POST_SETS."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import POST_SETS, post_test_accum
    from ref_gpu import RefPost
    rows = 4
    band = post_test_accum(rows)
    rp = RefPost()
    outs = {f"out{k}": rp.run(band, *ps)[:, :3].copy() for k, ps in enumerate(POST_SETS)}
    rp.close()
    path = os.path.join(outdir, "refpost.npz")
    np.savez_compressed(path, rows=np.int32(rows), params=np.array(POST_SETS, np.float64), **outs)
    print("post ->", path, os.path.getsize(path) // 1024, "KiB", flush=True)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden")
    os.makedirs(out, exist_ok=True)
    only = sys.argv[2:]
    for name in CASES:
        if not only or name in only or "stages" in only:
            make(name, out)
    for name in FRAME_CASES:
        if not only or name in only or "frames" in only:
            make_frame(name, out)
    if not only or "post" in only:
        make_post(out)
