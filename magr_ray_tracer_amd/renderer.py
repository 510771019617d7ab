"""Device: Python face of one librt355.so context (one per GPU), i.e. the launch/upload half
of the reference's Renderer (src/renderer.cpp:64-94,142-263,289-301) behind the C-ABI of
include/rt355.h.  Everything that computes runs in the HIP kernels; this file only moves
pointers.  `Renderer` wraps the C++ Renderer mirror (Init / Tick / FocusCamera / energy).
"""
import ctypes as C

import numpy as np

from . import _lib


class RtError(RuntimeError):
    pass


class Device:
    def __init__(self, width, height, y0=0, y1=None, shading=_lib.SHADING_NEE, sampling=_lib.SAMPLING_COSINE,
                 accel=_lib.ACCEL_BVH2, russian_roulette=True, filter_fireflies=True, max_bounces=_lib.MAX_BOUNCES,
                 device=0, profile=False, extend_variant=0, shade_blocks_per_cu=0, persist_blocks_per_cu=0, lib=None):
        self._lib = _lib.device_lib(lib)      # lib="refb": the build with the reference's OpenCL builtin sequences (tests only)
        cfg = np.zeros((), dtype=_lib.Config)
        cfg["width"], cfg["height"], cfg["y0"], cfg["y1"] = width, height, y0, height if y1 is None else y1
        cfg["max_bounces"], cfg["shading"], cfg["sampling"], cfg["accel"] = max_bounces, shading, sampling, accel
        cfg["russian_roulette"], cfg["filter_fireflies"] = int(russian_roulette), int(filter_fireflies)
        cfg["device"], cfg["profile"], cfg["extend_variant"] = device, (2 if profile is True else int(profile)), extend_variant
        cfg["shade_blocks_per_cu"], cfg["persist_blocks_per_cu"] = shade_blocks_per_cu, persist_blocks_per_cu
        self.cfg = cfg
        self.width, self.height = width, height
        self.y0, self.y1 = int(cfg["y0"]), int(cfg["y1"])
        self.npix = (self.y1 - self.y0) * width
        self.first_pixel = self.y0 * width
        self.accel = accel
        h = C.c_void_p()
        self._h = None
        self._chk(self._lib.rt_create(cfg.ctypes.data_as(C.c_void_p), C.byref(h)))
        self._h = h
        self._keep = None

    @classmethod
    def borrowed(cls, handle, cfg):
        """A Device view of a context somebody else owns (a lane of a Group): same methods, close() does not destroy it."""
        d = cls.__new__(cls)
        d._lib = _lib.device_lib()
        d.cfg = cfg
        d.width, d.height = int(cfg["width"]), int(cfg["height"])
        d.y0, d.y1 = int(cfg["y0"]), int(cfg["y1"])
        d.npix = (d.y1 - d.y0) * d.width
        d.first_pixel = d.y0 * d.width
        d.accel = int(cfg["accel"])
        d._h, d._keep, d._owned = C.c_void_p(handle), None, False
        return d

    def _chk(self, rc):
        if rc != 0:
            raise RtError(self._lib.rt_last_error().decode())

    def close(self):
        if self._h and getattr(self, "_owned", True):
            self._lib.rt_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- uploads (renderer.cpp:160-208)
    def upload(self, sa):
        nodes = sa.nodes(self.accel)
        P = _lib.ptr
        self._chk(self._lib.rt_upload_scene(
            self._h, P(sa.prims), len(sa.prims), P(sa.mats), len(sa.mats), P(sa.tex) if len(sa.tex) else None, len(sa.tex),
            P(sa.lights) if len(sa.lights) else None, len(sa.lights), P(nodes), len(nodes), P(sa.primIdx), len(sa.primIdx),
            P(sa.tlas), len(sa.tlas), P(sa.blas), len(sa.blas)))

    def share_scene(self, other):
        """Render the scene `other` (a Device on the same GPU, same accel) holds, from ITS device copy (rt_share_scene)."""
        self._chk(self._lib.rt_share_scene(self._h, other._h))

    def kernel_info(self):
        """Which traversal kernels this context runs for the uploaded scene (rt_kernel_info)."""
        k = np.zeros((), dtype=_lib.KernelInfo)
        self._chk(self._lib.rt_kernel_info(self._h, k.ctypes.data_as(C.c_void_p)))
        return {n: int(k[n]) for n in k.dtype.names}

    def extend_kernel_name(self):
        k = self.kernel_info()
        if k["persist4"]:
            return "k_trace_persist4<false>"
        if k["persist"] == 3:
            return "k_trace_persist_tlas<false, false, true> (LDS stack of %d entries per lane, deeper entries in global memory)" % k["stack_entries"]
        if k["persist"] == 2:
            return "k_trace_persist_tlas<false> (bounce 0: its one-ray-per-lane branch)"
        if k["persist"]:
            return "k_trace_persist<false> (bounce 0: <false, true>, node records of wave-uniform visits through the scalar cache)"
        return "k_extend<%s, %d>" % ("RT_ACCEL_BVH4" if self.accel == _lib.ACCEL_BVH4 else "RT_ACCEL_BVH2", k["layout"])

    def set_seeds(self, seeds):
        s = np.ascontiguousarray(seeds, dtype=np.uint32)
        self._chk(self._lib.rt_set_seeds(self._h, _lib.ptr(s), s.size))

    def seed_default(self):
        self._chk(self._lib.rt_seed_default(self._h))

    def get_seeds(self):
        s = np.zeros(self.npix, dtype=np.uint32)
        self._chk(self._lib.rt_get_seeds(self._h, _lib.ptr(s), s.size))
        return s

    def bind_accum(self, tensor):
        """Render into a torch CUDA tensor of shape (H, W, 4) float32 (kept alive by this object)."""
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() == self.width * self.height * 4
        self._keep = tensor
        self._chk(self._lib.rt_bind_accum(self._h, C.c_void_p(tensor.data_ptr())))

    # ---- frame (renderer.cpp:26-94)
    def reset(self):
        self._chk(self._lib.rt_reset(self._h))

    def render(self, cam, frames=1, antiAliasing=1, renderBVH=0):
        s = np.zeros((), dtype=_lib.Settings)
        s["antiAliasing"], s["renderBVH"] = antiAliasing, renderBVH
        c = np.ascontiguousarray(cam)
        self._chk(self._lib.rt_render(self._h, c.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), int(frames)))

    def synchronize(self):
        self._chk(self._lib.rt_synchronize(self._h))

    def focus(self, x, y, cam):
        t = C.c_float(0)
        c = np.ascontiguousarray(cam)
        self._chk(self._lib.rt_focus(self._h, int(x), int(y), c.ctypes.data_as(C.c_void_p), C.byref(t)))
        return np.float32(t.value)

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._chk(self._lib.rt_read_accum(self._h, _lib.ptr(out)))
        return out

    def write_accum(self, accum):
        a = np.ascontiguousarray(accum, dtype=np.float32)
        assert a.shape == (self.height, self.width, 4)
        self._chk(self._lib.rt_write_accum(self._h, _lib.ptr(a)))

    def save_checkpoint(self, path, frames):
        """{accum, seeds, frames}: everything a render carries from one frame to the next."""
        np.savez_compressed(path, accum=self.read_accum(), seeds=self.get_seeds(), frames=np.int32(frames),
                            dims=np.array([self.width, self.height, self.y0, self.y1], np.int32))

    def load_checkpoint(self, path):
        g = np.load(path)
        assert tuple(g["dims"]) == (self.width, self.height, self.y0, self.y1), "checkpoint was taken with another frame/band"
        self.write_accum(g["accum"])
        self.set_seeds(g["seeds"])
        return int(g["frames"])

    def postproc(self, frames, vignette=0.0, gamma=0.9, chromatic=0.0):
        """Renderer::PostProc + SaveFrame: returns (float image (H,W,4), RGBA8 image (H,W,4) uint8)."""
        f = np.zeros((self.height, self.width, 4), dtype=np.float32)
        b = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._chk(self._lib.rt_postproc(self._h, int(frames), float(vignette), float(gamma), float(chromatic), _lib.ptr(f), _lib.ptr(b)))
        return f, b

    def counters(self):
        c = np.zeros((), dtype=_lib.Counters)
        self._chk(self._lib.rt_read_counters(self._h, c.ctypes.data_as(C.c_void_p)))
        return {k: int(c[k]) for k in c.dtype.names}

    def reset_counters(self):
        self._chk(self._lib.rt_reset_counters(self._h))

    def stage_times(self):
        t = np.zeros((), dtype=_lib.StageTimes)
        self._chk(self._lib.rt_read_stage_times(self._h, t.ctypes.data_as(C.c_void_p)))
        return {k: (float(t[k]) if k.endswith("_ms") else int(t[k])) for k in t.dtype.names}

    def set_profile(self, level):
        self._chk(self._lib.rt_set_profile(self._h, int(level)))

    def reset_stage_times(self):
        self._chk(self._lib.rt_reset_stage_times(self._h))

    # ---- single stages (parity tests)
    def stage_begin_frame(self):
        self._chk(self._lib.rt_stage_begin_frame(self._h))

    def stage_generate(self, cam, antiAliasing=1):
        s = np.zeros((), dtype=_lib.Settings)
        s["antiAliasing"] = antiAliasing
        c = np.ascontiguousarray(cam)
        self._chk(self._lib.rt_stage_generate(self._h, c.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p)))

    def stage_extend(self, bounce, renderBVH=0):
        self._chk(self._lib.rt_stage_extend(self._h, bounce, renderBVH))

    def stage_shade(self, bounce):
        self._chk(self._lib.rt_stage_shade(self._h, bounce))

    def stage_connect(self, b0, b1):
        self._chk(self._lib.rt_stage_connect(self._h, b0, b1))

    def get_rays(self, bounce):
        n = C.c_int32(0)
        self._chk(self._lib.rt_debug_get_rays(self._h, bounce, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=_lib.Ray)
        if n.value:
            self._chk(self._lib.rt_debug_get_rays(self._h, bounce, _lib.ptr(out), n.value, C.byref(n)))
        return out

    def set_rays(self, bounce, rays):
        r = np.ascontiguousarray(rays, dtype=_lib.Ray)
        self._chk(self._lib.rt_debug_set_rays(self._h, bounce, _lib.ptr(r) if len(r) else _lib.ptr(np.zeros(1, dtype=_lib.Ray)), len(r)))

    def get_shadow(self, b0, b1):
        n = C.c_int32(0)
        self._chk(self._lib.rt_debug_get_shadow(self._h, b0, b1, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=_lib.ShadowRecord)
        if n.value:
            self._chk(self._lib.rt_debug_get_shadow(self._h, b0, b1, _lib.ptr(out), n.value, C.byref(n)))
        return out

    def enable_steps(self, on=True):
        self._chk(self._lib.rt_debug_enable_steps(self._h, int(on)))

    def get_steps(self):
        n = C.c_int32(0)
        out = np.zeros(self.npix, dtype=np.int32)
        self._chk(self._lib.rt_debug_get_steps(self._h, _lib.ptr(out), out.size, C.byref(n)))
        return out


class Group:
    """One accumulation rendered as `lanes` interleaved sample streams behind one handle (rt_group_*, include/rt355.h): own context,
    stream, queues and seed slice per lane, ONE device copy of the scene, accumulator = sum of the lanes in lane order."""

    def __init__(self, width, height, lanes=4, y0=0, y1=None, shading=_lib.SHADING_NEE, sampling=_lib.SAMPLING_COSINE,
                 accel=_lib.ACCEL_BVH2, russian_roulette=True, filter_fireflies=True, max_bounces=_lib.MAX_BOUNCES,
                 device=0, profile=False, extend_variant=0, shade_blocks_per_cu=0, persist_blocks_per_cu=0):
        self._lib = _lib.device_lib()
        cfg = np.zeros((), dtype=_lib.Config)
        cfg["width"], cfg["height"], cfg["y0"], cfg["y1"] = width, height, y0, height if y1 is None else y1
        cfg["max_bounces"], cfg["shading"], cfg["sampling"], cfg["accel"] = max_bounces, shading, sampling, accel
        cfg["russian_roulette"], cfg["filter_fireflies"] = int(russian_roulette), int(filter_fireflies)
        cfg["device"], cfg["profile"], cfg["extend_variant"] = device, (2 if profile is True else int(profile)), extend_variant
        cfg["shade_blocks_per_cu"], cfg["persist_blocks_per_cu"] = shade_blocks_per_cu, persist_blocks_per_cu
        self.cfg, self.width, self.height, self.accel = cfg, width, height, accel
        self.y0, self.y1 = int(cfg["y0"]), int(cfg["y1"])
        h = C.c_void_p()
        self._h = None
        self._chk(self._lib.rt_group_create(cfg.ctypes.data_as(C.c_void_p), int(lanes), C.byref(h)))
        self._h = h
        self.devs = [Device.borrowed(self._lib.rt_group_lane(self._h, m), cfg) for m in range(lanes)]

    def _chk(self, rc):
        if rc != 0:
            raise RtError(self._lib.rt_last_error().decode())

    def __len__(self):
        return len(self.devs)

    def concurrency(self):
        """How many of the lanes' HIP streams were measured to run side by side when the group was created."""
        return int(self._lib.rt_group_concurrency(self._h))

    def frames(self):
        return int(self._lib.rt_group_frames(self._h))

    def upload(self, sa):
        nodes = sa.nodes(self.accel)
        P = _lib.ptr
        self._chk(self._lib.rt_group_upload_scene(
            self._h, P(sa.prims), len(sa.prims), P(sa.mats), len(sa.mats), P(sa.tex) if len(sa.tex) else None, len(sa.tex),
            P(sa.lights) if len(sa.lights) else None, len(sa.lights), P(nodes), len(nodes), P(sa.primIdx), len(sa.primIdx),
            P(sa.tlas), len(sa.tlas), P(sa.blas), len(sa.blas)))

    def share_scene(self, other):
        """Render the scene another Group on the same GPU holds, from its device copy."""
        self._chk(self._lib.rt_group_share_scene(self._h, other._h))

    def seed(self, first_stream=0):
        self._chk(self._lib.rt_group_seed(self._h, int(first_stream)))

    def reset(self):
        self._chk(self._lib.rt_group_reset(self._h))

    def render(self, cam, frames=1, antiAliasing=1):
        s = np.zeros((), dtype=_lib.Settings)
        s["antiAliasing"] = antiAliasing
        c = np.ascontiguousarray(cam)
        self._chk(self._lib.rt_group_render(self._h, c.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), int(frames)))

    def synchronize(self):
        self._chk(self._lib.rt_group_synchronize(self._h))

    def sum_into(self, tensor):
        """Lane-ordered sum of the lanes' accumulators into a torch CUDA tensor (H, W, 4) float32, queued behind the pending frames."""
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() == self.width * self.height * 4
        self._chk(self._lib.rt_group_sum(self._h, C.c_void_p(tensor.data_ptr())))

    def focus(self, x, y, cam):
        t = C.c_float(0)
        c = np.ascontiguousarray(cam)
        self._chk(self._lib.rt_group_focus(self._h, int(x), int(y), c.ctypes.data_as(C.c_void_p), C.byref(t)))
        return np.float32(t.value)

    def read_accum(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        self._chk(self._lib.rt_group_read_accum(self._h, _lib.ptr(out)))
        return out

    def postproc(self, frames=0, vignette=0.0, gamma=0.9, chromatic=0.0):
        f = np.zeros((self.height, self.width, 4), dtype=np.float32)
        b = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self._chk(self._lib.rt_group_postproc(self._h, int(frames), float(vignette), float(gamma), float(chromatic), _lib.ptr(f), _lib.ptr(b)))
        return f, b

    def counters(self):
        """Work totals over the lanes."""
        tot = {}
        for d in self.devs:
            for k, v in d.counters().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def close(self):
        if self._h:
            self.devs = []
            self._lib.rt_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """The C++ Renderer mirror (host/renderer.cpp): Init(), Tick(), accumulator read-back, energy."""

    def __init__(self, scene, width, height, device=0, y0=0, y1=-1, shading=_lib.SHADING_NEE, sampling=_lib.SAMPLING_COSINE,
                 bvh=_lib.ACCEL_BVH2, russian_roulette=True, filter_fireflies=True):
        self._lib = _lib.host_lib()
        self.width, self.height = width, height
        self._h = self._lib.rth_renderer_create(scene._h, width, height, device, y0, y1, shading, sampling, bvh,
                                                int(russian_roulette), int(filter_fireflies))
        if not self._h:
            raise RtError(self._lib.rth_last_error().decode())

    def _chk(self, rc):
        if rc < 0:
            raise RtError(self._lib.rth_last_error().decode())

    def SetCamera(self, origin, forward, fov=110.0, aperture=0.1):
        self._chk(self._lib.rth_renderer_set_camera(self._h, _lib.fvec(origin), _lib.fvec(forward), float(fov), float(aperture)))

    def SetLanes(self, lanes):
        """Before Init(): render the accumulation as `lanes` interleaved sample streams; a Tick() is then `lanes` frames."""
        self._chk(self._lib.rth_renderer_set_lanes(self._h, int(lanes)))

    def Init(self):
        self._chk(self._lib.rth_renderer_init(self._h))

    def Tick(self, frames=1):
        self._chk(self._lib.rth_renderer_tick(self._h, int(frames)))

    def camera(self):
        cam = np.zeros((), dtype=_lib.Camera)
        self._chk(self._lib.rth_renderer_camera(self._h, cam.ctypes.data_as(C.c_void_p)))
        return cam

    def read(self):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        e = C.c_float(0)
        self._chk(self._lib.rth_renderer_read(self._h, _lib.ptr(out), C.byref(e)))
        return out, float(e.value)

    def Move(self, camdir):
        self._chk(self._lib.rth_renderer_camera_move(self._h, int(camdir)))

    def MouseMove(self, dx, dy):
        self._chk(self._lib.rth_renderer_camera_mouse(self._h, float(dx), float(dy)))

    def Zoom(self, offset):
        self._chk(self._lib.rth_renderer_camera_zoom(self._h, float(offset)))

    def frames(self):
        return int(self._lib.rth_renderer_frames(self._h))

    def SaveFrame(self, path):
        self._chk(self._lib.rth_renderer_save_frame(self._h, str(path).encode()))

    def close(self):
        if self._h:
            self._lib.rth_renderer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
