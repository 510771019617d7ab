"""Repeats test_two_processes_with_two_lanes_each_share_the_gpu (two processes x two contexts on one GPU, 48 frames of the bench scene each)
and prints every outcome; exits non-zero on the first mismatch or device fault."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
from test_gpu_parity import _render_lanes_crc

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        solo = pool.map(_render_lanes_crc, [(2, 48)])[0]
    print("solo", solo, flush=True)
    for i in range(n):
        t = time.time()
        with ctx.Pool(2) as pool:
            both = pool.map(_render_lanes_crc, [(2, 48), (2, 48)], chunksize=1)
        print(i, both, "ok" if both == [solo, solo] else "MISMATCH", f"{time.time() - t:.1f}s", flush=True)
        if both != [solo, solo]:
            sys.exit(1)
