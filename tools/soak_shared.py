"""One-off soak: N processes render the same frames on ONE GPU at the same time; all must finish without a device fault and agree bit for
bit with each other (k_shade's scan may not depend on residency; the traversal kernels have no inter-workgroup dependency)."""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def work(args):
    frames, W, H = args
    import numpy as np
    from magr_ray_tracer_amd import scenes
    from magr_ray_tracer_amd.renderer import Device
    s, view = scenes.sponza_class(float(os.environ.get("SOAK_DETAIL", "0.5")))
    d = Device(W, H)
    d.upload(s.arrays())
    d.seed_default()
    t = time.time()
    d.render(scenes.camera_for(view, W, H), frames)
    d.synchronize()
    dt = time.time() - t
    out = (int(d.read_accum().view(np.uint32).astype(np.uint64).sum()), int(d.get_seeds().astype(np.uint64).sum()), round(dt, 2))
    d.close()
    return out


if __name__ == "__main__":
    procs, frames = int(sys.argv[1]), int(sys.argv[2])
    W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1280, 720)
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        res = pool.map(work, [(frames, W, H)] * procs, chunksize=1)
    print("results:", res)
    ok = all(r[:2] == res[0][:2] for r in res)
    print("soak_shared:", procs, "processes x", frames, "frames", "identical" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
