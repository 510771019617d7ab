"""ctypes face of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (magr_ray_tracer_amd) never does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
S1, S0 = 1, 0

OrcConfig = np.dtype([(n, "<i4") for n in ("width", "height", "max_bounces", "shading", "sampling", "accel",
                                            "russian_roulette", "filter_fireflies", "schedule")])
OrcCounters = np.dtype([(n, "<u8") for n in ("rays", "tlas_visits", "inst_visits", "node_visits", "prim_tests")])


class _Scene(C.Structure):
    _fields_ = [("prims", C.c_void_p), ("nPrims", C.c_int32), ("mats", C.c_void_p), ("nMats", C.c_int32),
                ("tex", C.c_void_p), ("nTex", C.c_int32), ("lights", C.c_void_p), ("nLights", C.c_int32),
                ("bvh2", C.c_void_p), ("bvh4", C.c_void_p), ("nNodes", C.c_int32), ("primIdx", C.c_void_p), ("nIdx", C.c_int32),
                ("tlas", C.c_void_p), ("nTlas", C.c_int32), ("blas", C.c_void_p), ("nBlas", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -m magr_ray_tracer_amd.build`")
        L = C.CDLL(path)
        vp, i32 = C.c_void_p, C.c_int32
        L.orc_seed_stream.argtypes = [vp, C.c_int64, C.c_int64]
        L.orc_generate.argtypes = [vp, i32, i32, vp, vp, i32, vp]
        L.orc_extend.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp]
        L.orc_shade.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_connect.argtypes = [vp, i32, vp, vp, vp, vp]
        L.orc_focus.argtypes = [i32, i32, vp, vp, vp]
        L.orc_focus.restype = C.c_float
        L.orc_frame_work_bytes.argtypes = [i32, vp]
        L.orc_frame_work_bytes.restype = C.c_size_t
        L.orc_render_frame.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
        L.orc_render_bands.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        L.orc_trace_normals.argtypes = [vp, vp, vp, i32, vp]
        L.orc_postproc.argtypes = [vp, i32, i32, i32, C.c_float, C.c_float, C.c_float, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


def seed_stream(first, n):
    s = np.zeros(n, dtype=np.uint32)
    lib().orc_seed_stream(_p(s), first, n)
    return s


def postproc(accum, frames, vignette=0.0, gamma=0.9, chromatic=0.0):
    a = np.ascontiguousarray(accum, dtype=np.float32)
    H, W = a.shape[:2]
    f = np.zeros((H, W, 4), np.float32)
    b = np.zeros((H, W, 4), np.uint8)
    lib().orc_postproc(_p(a), W, H, int(frames), float(vignette), float(gamma), float(chromatic), _p(f), _p(b))
    return f, b


class Oracle:
    """Holds one scene (SceneArrays-like object with numpy arrays) + variant config."""

    def __init__(self, sa, width, height, shading=1, sampling=1, accel=0, russian_roulette=True, filter_fireflies=True,
                 max_bounces=7, schedule=S1):
        self.sa = sa
        self.cfg = np.zeros((), dtype=OrcConfig)
        c = self.cfg
        c["width"], c["height"], c["max_bounces"] = width, height, max_bounces
        c["shading"], c["sampling"], c["accel"] = shading, sampling, accel
        c["russian_roulette"], c["filter_fireflies"], c["schedule"] = int(russian_roulette), int(filter_fireflies), schedule
        self.width, self.height = width, height
        s = _Scene()
        s.prims, s.nPrims = _p(sa.prims), len(sa.prims)
        s.mats, s.nMats = _p(sa.mats), len(sa.mats)
        s.tex, s.nTex = _p(sa.tex), len(sa.tex)
        s.lights, s.nLights = _p(sa.lights), len(sa.lights)
        s.bvh2, s.bvh4 = _p(sa.bvh2), _p(sa.bvh4)
        s.nNodes = len(sa.bvh2)
        s.primIdx, s.nIdx = _p(sa.primIdx), len(sa.primIdx)
        s.tlas, s.nTlas = _p(sa.tlas), len(sa.tlas)
        s.blas, s.nBlas = _p(sa.blas), len(sa.blas)
        self._scene = s

    @property
    def _sc(self):
        return C.cast(C.pointer(self._scene), C.c_void_p)

    @property
    def _cfg(self):
        return self.cfg.ctypes.data_as(C.c_void_p)

    def generate(self, cam, first_pixel, n, seeds, antiAliasing=1):
        from magr_ray_tracer_amd import _lib as W
        rays = np.zeros(n, dtype=W.Ray)
        c = np.ascontiguousarray(cam)
        lib().orc_generate(_p(rays), n, first_pixel, self._cfg, c.ctypes.data_as(C.c_void_p), antiAliasing, _p(seeds))
        return rays

    def extend(self, rays, renderBVH=0, accum=None, want_steps=False):
        steps = np.zeros(len(rays), dtype=np.int32) if want_steps else None
        ctr = np.zeros((), dtype=OrcCounters)
        lib().orc_extend(_p(rays), len(rays), self._sc, self._cfg, renderBVH, _p(accum) if accum is not None else None,
                         _p(steps) if steps is not None else None, ctr.ctypes.data_as(C.c_void_p))
        return steps, {k: int(ctr[k]) for k in ctr.dtype.names}

    def shade(self, rays, accum, seeds, shadow_capacity=None):
        from magr_ray_tracer_amd import _lib as W
        n = len(rays)
        out = np.zeros(max(n, 1), dtype=W.Ray)
        sh = np.zeros(max(shadow_capacity or n, 1), dtype=W.ShadowRay)
        nOut, nSh = C.c_int32(0), C.c_int32(0)
        lib().orc_shade(_p(rays), n, _p(out), C.addressof(nOut), _p(sh), C.addressof(nSh), self._sc, self._cfg, _p(accum), _p(seeds))
        return out[:nOut.value].copy(), sh[:nSh.value].copy()

    def connect(self, shadow, accum):
        ctr = np.zeros((), dtype=OrcCounters)
        if len(shadow):
            lib().orc_connect(_p(shadow), len(shadow), self._sc, self._cfg, _p(accum), ctr.ctypes.data_as(C.c_void_p))
        return {k: int(ctr[k]) for k in ctr.dtype.names}

    def focus(self, x, y, cam):
        c = np.ascontiguousarray(cam)
        return np.float32(lib().orc_focus(x, y, self._sc, self._cfg, c.ctypes.data_as(C.c_void_p)))

    def render(self, cam, frames, accum=None, seeds=None, y0=0, y1=None, threads=1, antiAliasing=1):
        """frames x RayTrace() over rows [y0,y1); threads>1 splits into independent row bands (baseline mode)."""
        y1 = self.height if y1 is None else y1
        n = (y1 - y0) * self.width
        if accum is None:
            accum = np.zeros((self.height, self.width, 4), dtype=np.float32)
        if seeds is None:
            seeds = seed_stream(y0 * self.width, n)
        e, c = np.zeros((), dtype=OrcCounters), np.zeros((), dtype=OrcCounters)
        cm = np.ascontiguousarray(cam)
        lib().orc_render_bands(self._sc, self._cfg, cm.ctypes.data_as(C.c_void_p), antiAliasing, y0, y1, frames, threads,
                               _p(accum), _p(seeds), e.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
        return accum, seeds, {k: int(e[k]) for k in e.dtype.names}, {k: int(c[k]) for k in c.dtype.names}

    def trace_normals(self, cam, threads=1):
        out = np.zeros((self.height, self.width, 4), dtype=np.float32)
        cm = np.ascontiguousarray(cam)
        lib().orc_trace_normals(self._sc, self._cfg, cm.ctypes.data_as(C.c_void_p), threads, _p(out))
        return out
