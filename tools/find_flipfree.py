"""GPU box: for each whole-frame case, which one-row bands of the 1280x720 frame does the free-running oracle (S0) follow the
reference's own kernels through all 7 bounces without meeting a knife-edge decision?  (tests/test_gpu_reference.py FLIPFREE)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import compare_frames_s0, oracle_frame_s0  # noqa: E402
from oracle.oracle_py import Oracle, S0  # noqa: E402
import test_gpu_reference as T  # noqa: E402

if __name__ == "__main__":
    for case in (sys.argv[1:] or list(T.FRAME_VARIANTS)):
        good = []
        _, _, (c0, c1), _ = T.FRAME_VARIANTS[case]
        for y in range(c0 - 4, c1 + 4):
            fn, v, sa, cam, cap, _ = T._reference_frame(case, (y, y + 1))
            o = Oracle(sa, T.RW, T.RH, **v, schedule=S0)
            try:
                st = compare_frames_s0(cap, oracle_frame_s0(o, cam, y, y + 1), case)
                good.append(y)
                print(case, y, "ok", st, flush=True)
            except AssertionError as e:
                print(case, y, "flip:", str(e).splitlines()[0][:160], flush=True)
        print(f'FLIPFREE "{case}": {[(y, y + 1) for y in good]}', flush=True)
