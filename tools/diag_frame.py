"""Diagnostic (GPU box): where does the oracle's S0 shade first part from the reference's own kernels?  Teacher-forced: at every
bounce the oracle shades the REFERENCE's rays, one ray at a time in S0 order, from the reference's RNG state before that launch;
the first ray whose outcome (survives / pixel / flags / direction) differs from the reference's next appended ray is printed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_gpu  # noqa: E402
from helpers import DEFAULT  # noqa: E402
from magr_ray_tracer_amd import scenes  # noqa: E402
from oracle.oracle_py import Oracle, S0  # noqa: E402
from test_gpu_reference import FRAME_VARIANTS  # noqa: E402

RW, RH = ref_gpu.REF_W, ref_gpu.REF_H


def main(case):
    fn, vo, (y0, y1), vi = FRAME_VARIANTS[case]
    v = dict(DEFAULT, **vi)
    s, view = fn()
    sa = s.arrays()
    cam = scenes.camera_for(dict(view, **vo), RW, RH)
    ref = ref_gpu.RefGPU(sa, **v)
    cam["focalLength"] = ref.focus(RW // 2, (y0 + y1) // 2, cam)
    cap = ref.frame_s0(cam, y0, y1, shading=v["shading"], russian_roulette=v["russian_roulette"])
    ref.close()
    o = Oracle(sa, RW, RH, **v, schedule=S0)
    acc = np.zeros((RH * RW, 4), np.float32)
    for b in range(7):
        rays = cap["ext"][b]
        nxt = cap["ext"][b + 1] if b + 1 < 7 else cap["last_out"]
        seed = np.array([cap["gen_seeds"][0] if b == 0 else cap["seed0"][b - 1]], np.uint32)
        j = 0
        bad = None
        for idx in range(len(rays) - 1, -1, -1):
            one = rays[idx:idx + 1].copy()
            out, sh = o.shade(one, acc, seed)
            if len(out):
                ok = j < len(nxt) and out["pixelIdx"][0] == nxt["pixelIdx"][j] and out["bounces"][0] == nxt["bounces"][j] and \
                    out["inside"][0] == nxt["inside"][j] and out["lastSpecular"][0] == nxt["lastSpecular"][j] and \
                    np.abs(out["D"][0] - nxt["D"][j]).max() < 1e-4
                if not ok:
                    bad = (idx, j, "oracle appended a ray the reference did not (or another one)")
                    break
                j += 1
            else:
                # the reference must not have appended this pixel here
                if j < len(nxt) and nxt["pixelIdx"][j] == one["pixelIdx"][0] and (idx == 0 or rays["pixelIdx"][idx - 1] != one["pixelIdx"][0]):
                    bad = (idx, j, "reference appended a ray, oracle killed the path")
                    break
        print(f"bounce {b}: {len(rays)} rays, oracle seed after {int(seed[0])} reference {cap['seed0'][b]}", "OK" if bad is None and int(seed[0]) == cap["seed0"][b] else "DIVERGED")
        if bad is not None:
            idx, j, why = bad
            r = rays[idx]
            p = sa.prims[r["primIdx"]] if r["primIdx"] >= 0 else None
            print("  first difference at slot", idx, "out index", j, ":", why)
            print("  ray:", {k: r[k].tolist() if hasattr(r[k], "tolist") else r[k] for k in r.dtype.names if not k.startswith("_")})
            if p is not None:
                m = sa.mats[p["matIdx"]]
                print("  prim type", int(p["objType"]), "mat", {k: m[k].tolist() for k in m.dtype.names if not k.startswith("_")})
            if j < len(nxt):
                print("  reference next out:", {k: nxt[j][k].tolist() for k in ("O", "D", "intensity", "pixelIdx", "bounces", "inside", "lastSpecular")})
            one = rays[idx:idx + 1].copy()
            print("  oracle out:", [{k: x[k].tolist() for k in ("O", "D", "intensity", "pixelIdx", "bounces", "inside", "lastSpecular")} for x in out])
            break


if __name__ == "__main__":
    for c in (sys.argv[1:] or list(FRAME_VARIANTS)):
        print("==", c, flush=True)
        main(c)
