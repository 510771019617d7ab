"""GPU box: one-row bands on which the HIP path, running a whole frame freely, makes every discrete decision the reference's own kernels
make under schedule S1 (RefGPU.frame_s1) - identical queue lengths per bounce and identical per-slot RNG states at the end."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_gpu  # noqa: E402
import test_gpu_reference as T  # noqa: E402
from magr_ray_tracer_amd.renderer import Device  # noqa: E402
from oracle.oracle_py import seed_stream  # noqa: E402


LIB = None      # "refb": the build with the reference's OpenCL builtin sequences (librt355_refb.so)


def run(case, y):
    fn, vo, _, vi = T.FRAME_VARIANTS[case]
    v = dict(T.DEFAULT, **vi)
    s, view = fn()
    sa = s.arrays()
    cam = T.scenes.camera_for(dict(view, **vo), T.RW, T.RH)
    ref = ref_gpu.RefGPU(sa, **v)
    cam["focalLength"] = ref.focus(T.RW // 2, y, cam)
    r = ref.frame_s1(cam, y, y + 1, shading=v["shading"], russian_roulette=v["russian_roulette"])
    ref.close()
    d = Device(T.RW, T.RH, y0=y, y1=y + 1, lib=LIB, **v)
    d.upload(sa)
    d.set_seeds(seed_stream(y * T.RW, T.RW))
    d.render(cam, 1)
    got = d.read_accum().reshape(-1, 4)[y * T.RW:(y + 1) * T.RW]
    seeds = d.get_seeds()
    counts = [len(d.get_rays(b)) for b in range(7)]
    d.close()
    rel = (np.abs(got.astype(np.float64) - r["accum"]) / np.maximum(np.abs(r["accum"]), 1e-3)).max(1)
    return counts == r["n_in"], bool(np.array_equal(seeds, r["seeds"])), float(rel.max()), int((rel > 1e-4).sum()), counts, r["n_in"]


if __name__ == "__main__":
    if "--refb" in sys.argv:
        sys.argv.remove("--refb")
        LIB = "refb"
        print("library: librt355_refb.so (-DRT355_REF_BUILTINS)")
    for case in (sys.argv[1:] or ["nee", "kajiya_hemi_norr", "fisheye", "nee_bvh4"]):
        good = []
        for y in range(352, 368):
            c_ok, s_ok, mx, bad, counts, rc = run(case, y)
            print(case, y, "counts", c_ok, "seeds", s_ok, "accum max rel", mx, "pixels > 1e-4:", bad, flush=True)
            if not c_ok:
                print("   queue lengths HIP", counts, "reference", rc, flush=True)
            if c_ok and s_ok:
                good.append(y)
        print(f'S1FREE "{case}": {good}', flush=True)
