"""One-off deep fuzz on the GPU box: the randomized parity test of tests/test_gpu_parity.py over many more seeds."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
big = "big" in sys.argv[3:]
mixed = "mixed" in sys.argv[3:]        # spheres, glass, textures, sphere lights, fisheye (tests/test_gpu_parity.py test_fuzz_random_mixed_scenes)
multi = "multi" in sys.argv[3:]        # 2-6 BLAS under a TLAS, random instance transforms, random path through k_trace_persist_tlas (test_fuzz_random_multi_blas_soups)


class _Env:
    """stand-in for pytest's monkeypatch: the multi-BLAS test picks its kernel path through environment variables"""
    def setenv(self, k, v):
        os.environ[k] = v

if big:
    os.environ["RT355_TUNE"] = "64,20,6,8,1"   # one workgroup per CU: queues above 65,536 rays take the persistent branch
bad = []
t = time.time()
for seed in range(first, first + count):
    try:
        if multi:
            for k in ("RT355_SPILL_CAP", "RT355_NO_SPILL", "RT355_TLAS_FLAT"):
                os.environ.pop(k, None)
            T.test_fuzz_random_multi_blas_soups(seed, _Env(), big)
        else:
            (T.test_fuzz_random_mixed_scenes if mixed else T.test_fuzz_random_triangle_soups)(seed, big)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
        print("MISMATCH seed", seed, str(e)[:300], flush=True)
    except Exception as e:   # builder / upload errors are findings too
        bad.append((seed, repr(e)[:200]))
        print("ERROR seed", seed, repr(e)[:300], flush=True)
    if (seed - first) % 50 == 49:
        print(f"  ... {seed - first + 1} scenes, {len(bad)} failures, {time.time() - t:.0f} s", flush=True)
print(f"deep fuzz: seeds {first}..{first + count - 1}, {len(bad)} failures, {time.time() - t:.1f} s")
sys.exit(1 if bad else 0)
