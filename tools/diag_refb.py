"""GPU box: where does the REF_BUILTINS build (librt355_refb.so) still differ from the reference's own shade kernel?  Per bounce of a
reference frame: number of rays whose O / D / intensity words differ, by material class of the hit, with the worst cases."""
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

case = sys.argv[1] if len(sys.argv) > 1 else "nee"
lib = None if "--ieee" in sys.argv else "refb"
fn, v, sa, cam, cap, (y0, y1) = T._reference_frame(case, (359, 361))
RW, RH = T.RW, T.RH
ref = ref_gpu.RefGPU(sa, **v)
d = Device(RW, RH, y0=y0, y1=y1, lib=lib, **v)
d.upload(sa)
first, n0 = y0 * RW, (y1 - y0) * RW
mats, prims = sa.mats, sa.prims
for b, ext in enumerate(cap["ext"]):
    n = len(ext)
    seeds = seed_stream(7919 * (b + 1), n0)
    ref.clear_accum()
    rout, rsh, rseeds = ref.shade_s1(ext, seeds[:n].copy())
    d.set_rays(b, ext)
    d.set_seeds(seeds)
    d.reset()
    d.stage_shade(b)
    out = d.get_rays(b + 1)
    sh = d.get_shadow(b, b)
    print(f"bounce {b}: {n} rays in, survivors HIP {len(out)} ref {len(rout)}, shadow HIP {len(sh)} ref {len(rsh)}, seeds equal {np.array_equal(d.get_seeds()[:n], rseeds[:n])}")
    if len(out) != len(rout):
        continue
    # map survivors back to their parent ray through the pixel index
    pix_in = {int(p): i for i, p in enumerate(ext["pixelIdx"])}
    for f in ("O", "D", "intensity"):
        a, r = out[f].view(np.uint32).reshape(len(out), -1), rout[f].view(np.uint32).reshape(len(out), -1)
        bad = np.nonzero((a != r).any(1))[0]
        if not len(bad):
            continue
        cls = {}
        for i in bad:
            src = ext[pix_in[int(out["pixelIdx"][i])]]
            m = mats[prims["matIdx"][src["primIdx"]]]
            key = ("glass" if m["isDielectric"] else ("spec%.2f" % m["specular"])) + (" tex" if m["texIdx"] != -1 else "") + (" inside" if src["inside"] else "") + f" type{int(prims['objType'][src['primIdx']])}" + (" lastSpec" if out["lastSpecular"][i] else "")
            cls[key] = cls.get(key, 0) + 1
        i = bad[0]
        print(f"   {f}: {len(bad)} of {len(out)} differ; by class {cls}; first: HIP {out[f][i]} ref {rout[f][i]}")
    if len(sh) == len(rsh) and len(sh):
        for f, g in (("tmax", None), ("radiance", None)):
            pass
        t_ref = rsh["dist"] - np.float32(2e-4)
        print("   shadow: tmax differ", int((sh["tmax"] != t_ref).sum()), "of", len(sh), " L differ", int((sh["l"] != rsh["L"][:, :3]).any(1).sum()),
              " origin differ", int((sh["o"] != (rsh["I"] + rsh["L"] * np.float32(1e-4))[:, :3]).any(1).sum()))
d.close()
ref.close()
