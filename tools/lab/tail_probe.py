"""lab: frames of the bench scene with a librt355.so built with -DRT355_TAIL_PROBE (tools/lab/tail_probe.sh): per persistent launch of the
LAST frame, when the waves found the queue dry and when they exited (per-wave records, read back through rt_lab_tail_probe)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from magr_ray_tracer_amd import scenes, _lib
from magr_ray_tracer_amd.renderer import Device
W, H = 1920, 1080
s, view = scenes.sponza_class(1.0)
sa = s.arrays()
cam = scenes.camera_for(view, W, H)
d = Device(W, H, profile=2)
d.upload(sa)
d.seed_default()
d.render(cam, int(sys.argv[1]) if len(sys.argv) > 1 else 40)
d.synchronize()
d.reset_stage_times()
t = time.perf_counter()
d.render(cam, 16)
d.synchronize()
ms = (time.perf_counter() - t) / 16 * 1e3
st = d.stage_times()
print(f"host: {ms:.3f} ms per frame; stage ms per frame: " + ", ".join(f"{k[:-3]} {st[k] / 16:.3f}" for k in st if k.endswith("_ms")))
buf = np.zeros((9, 8192, 4), np.uint64)
lib = _lib.device_lib()
assert lib.rt_lab_tail_probe(buf.ctypes.data_as(C.c_void_p)) == 0
TICK_US = float(os.environ.get("TICK_US", "0.01"))   # wall_clock64: 100 MHz
for slot in range(9):
    r = buf[slot]
    r = r[r[:, 2] > 0]
    if not len(r):
        continue
    t0 = r[:, 0].min()
    start = (r[:, 0] - t0) * TICK_US
    ex = (r[:, 2] - t0) * TICK_US
    dry = r[r[:, 1] > 0, 1]
    dry = (dry - t0) * TICK_US if len(dry) else np.zeros(1)
    life = ex - start
    pc = lambda a, q: float(np.percentile(a, q))
    name = f"extend bounce {slot}" if slot < 8 else "connect"
    print(f"{name}: waves {len(r)} rays {int(r[:, 3].sum())} | last start {start.max():.1f} | queue dry first {dry.min():.1f} median {pc(dry, 50):.1f} last {dry.max():.1f} | "
          f"exit p10 {pc(ex, 10):.1f} p50 {pc(ex, 50):.1f} p90 {pc(ex, 90):.1f} p99 {pc(ex, 99):.1f} last {ex.max():.1f} us | mean wave life {life.mean():.1f} us "
          f"= {life.mean() / ex.max():.2f} of the launch")
d.close()
