"""Diagnostic (GPU box): free-running oracle (S0) vs the reference's own kernels - how far apart are the rays entering bounce b,
and which rays land on another primitive?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_gpu  # noqa: E402
from helpers import DEFAULT, oracle_frame_s0  # noqa: E402
from magr_ray_tracer_amd import scenes  # noqa: E402
from oracle.oracle_py import Oracle, S0  # noqa: E402
from test_gpu_reference import FRAME_VARIANTS  # noqa: E402

RW, RH = ref_gpu.REF_W, ref_gpu.REF_H
np.set_printoptions(precision=9, linewidth=200)


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
    mine = oracle_frame_s0(o, cam, y0, y1)
    for b in range(7):
        r, m = cap["ext"][b], mine["ext"][b]
        print(f"bounce {b}: n_in {len(r)} / {len(m)}  seed0 {cap['seed0'][b]} / {mine['seed0'][b]}")
        if len(r) != len(m) or not np.array_equal(r["pixelIdx"], m["pixelIdx"]):
            print("  queues differ in length or pixel order: stop")
            break
        dD = np.abs(r["D"] - m["D"]).max(1) / np.maximum(np.abs(r["D"]).max(1), 1e-30)
        dO = np.abs(r["O"] - m["O"]).max(1)
        print("  max rel dD", dD.max(), "max abs dO", dO.max(), " rays with dD > 1e-5:", int((dD > 1e-5).sum()), " > 1e-6:", int((dD > 1e-6).sum()))
        bad = np.nonzero(r["primIdx"] != m["primIdx"])[0]
        print("  primIdx differs on", len(bad), "rays")
        for i in bad[:6]:
            print("   slot", i, "pixel", r["pixelIdx"][i], "prim ref/orc", r["primIdx"][i], m["primIdx"][i], "t", r["t"][i], m["t"][i], "bounces", r["bounces"][i], "inside", r["inside"][i])
            print("     O ref", r["O"][i], "\n     O orc", m["O"][i], "\n     D ref", r["D"][i], "\n     D orc", m["D"][i])
        big = np.argsort(-dD)[:3]
        for i in big:
            print("   largest dD: slot", i, "dD", dD[i], "D ref", r["D"][i], "orc", m["D"][i], "prim", r["primIdx"][i], "lastSpec", r["lastSpecular"][i])
        if len(bad):
            break


if __name__ == "__main__":
    for c in (sys.argv[1:] or ["nee"]):
        print("==", c, flush=True)
        main(c)
