"""procs x lanes contexts on ONE GPU: every process runs dist.Lanes with `lanes` contexts; all must finish without a device fault."""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def work(args):
    lanes, frames, W, H = args
    import numpy as np
    from magr_ray_tracer_amd import dist as rdist, scenes
    from magr_ray_tracer_amd.renderer import Device
    from oracle.oracle_py import seed_stream
    s, view = scenes.sponza_class(float(os.environ.get("SOAK_DETAIL", "1.0")))
    sa = s.arrays()

    def make(m):
        d = Device(W, H)
        d.upload(sa)
        return d
    g = rdist.Lanes(lanes, make, lambda m: seed_stream(m * W * H, W * H))
    t = time.time()
    try:
        g.render(scenes.camera_for(view, W, H), frames)
        g.synchronize()
    except Exception as e:
        return ("FAULT", str(e)[:120], round(time.time() - t, 2))
    out = (int(g.read_accum().view(np.uint32).astype(np.uint64).sum()), round(time.time() - t, 2))
    g.close()
    return out


if __name__ == "__main__":
    procs, lanes, frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        res = pool.map(work, [(lanes, frames, 1920, 1080)] * procs, chunksize=1)
    print(f"procs {procs} x lanes {lanes} x {frames} frames:", res, flush=True)
