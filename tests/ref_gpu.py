"""TEST INFRASTRUCTURE: drive the reference's own OpenCL kernels (oracle/_ref/*.co, built by
oracle/build_ref.sh from /root/reference, unmodified) on the GPU through oracle/_ref/libref_runner.so.
Used by tests/test_gpu_reference.py and tests/golden/make_golden.py.  Resolution is the reference's
compile-time 1280x720 (src/constants.h:3-4)."""
import ctypes as C
import os

import numpy as np

from magr_ray_tracer_amd import _lib as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
REF_W, REF_H = 1280, 720


def variant_file(shading=1, sampling=1, accel=0, russian_roulette=True, filter_fireflies=True):
    return "wf_%s_%s_%s_rr%d_ff%d.co" % ("nee" if shading else "simple", "cosine" if sampling else "hemisphere",
                                         "bvh4" if accel else "bvh2", int(russian_roulette), int(filter_fireflies))


def available():
    return os.path.exists(os.path.join(REF_DIR, "libref_runner.so")) and os.path.exists(os.path.join(REF_DIR, variant_file()))


class RefGPU:
    def __init__(self, sa, **variant):
        self.R = C.CDLL(os.path.join(REF_DIR, "libref_runner.so"))
        self.R.ref_last_error.restype = C.c_char_p
        self._chk(self.R.ref_init(0))
        self.mod = C.c_void_p()
        self._chk(self.R.ref_load(os.path.join(REF_DIR, variant_file(**variant)).encode(), C.byref(self.mod)))
        self.accel = variant.get("accel", 0)
        self.sa = sa
        self._bufs = []
        d = self.dbuf
        self.prims, self.mats = d(sa.prims), d(sa.mats)
        self.tex = d(sa.tex if len(sa.tex) else np.zeros(4, np.float32))
        self.lights = d(sa.lights if len(sa.lights) else np.zeros(1, np.uint32))
        self.tlas, self.blas, self.idx = d(sa.tlas), d(sa.blas), d(sa.primIdx)
        self.nodes = d(sa.bvh4 if self.accel else sa.bvh2)
        self.accum = d(nbytes=16 * REF_W * REF_H)
        self.settings = np.zeros(1, dtype=W.Settings)
        self.settings["antiAliasing"] = 1
        self.settings["numLights"] = len(sa.lights)
        self.settings["numPrimitives"] = len(sa.prims)
        self.d_set = d(self.settings)

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.R.ref_last_error().decode())

    def dbuf(self, arr=None, nbytes=None):
        p = C.c_void_p()
        nb = arr.nbytes if arr is not None else nbytes
        self._chk(self.R.ref_malloc(C.byref(p), C.c_size_t(nb)))
        if arr is not None and nb:
            a = np.ascontiguousarray(arr)
            self._chk(self.R.ref_h2d(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(nb)))
        self._bufs.append(p)
        return p

    def h2d(self, p, arr):
        a = np.ascontiguousarray(arr)
        self._chk(self.R.ref_h2d(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)))

    def rd(self, p, dtype, n):
        out = np.zeros(n, dtype=dtype)
        self._chk(self.R.ref_d2h(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(out.nbytes)))
        return out

    def launch(self, name, g, l, args):
        hold = [a if isinstance(a, np.ndarray) else C.c_void_p(a.value) for a in args]
        arr = (C.c_void_p * len(args))(*[h.ctypes.data_as(C.c_void_p) if isinstance(h, np.ndarray) else C.cast(C.pointer(h), C.c_void_p)
                                         for h in hold])
        self._chk(self.R.ref_launch(self.mod, name.encode(), g, l, arr))

    def launch_timed(self, name, g, l, args):
        hold = [a if isinstance(a, np.ndarray) else C.c_void_p(a.value) for a in args]
        arr = (C.c_void_p * len(args))(*[h.ctypes.data_as(C.c_void_p) if isinstance(h, np.ndarray) else C.cast(C.pointer(h), C.c_void_p)
                                         for h in hold])
        ms = C.c_float(0)
        self._chk(self.R.ref_launch_timed(self.mod, name.encode(), g, l, arr, C.byref(ms)))
        return ms.value

    def extend_timed(self, rays, global_size, local_size=256):
        """The reference's extend with `global_size` persistent work-items (reference: 2560, constants.h:28-31).
        Timing only: with more than one work-group the kernel's counter swap races (Appendix B #3), so numInRays is
        pre-set and the result is not used for parity."""
        n = len(rays)
        d_rays = self.dbuf(rays)
        self.set_counts(n, n, 0)
        return self.launch_timed("extend", global_size, local_size, [d_rays, self.prims, self.tlas, self.blas, self.nodes, self.idx, self.accum, self.d_set])

    def set_counts(self, numIn=0, numOut=0, shadow=0):
        self.settings["numInRays"], self.settings["numOutRays"], self.settings["shadowRays"] = numIn, numOut, shadow
        self.h2d(self.d_set, self.settings)

    def get_settings(self):
        return self.rd(self.d_set, W.Settings, 1)[0]

    def clear_accum(self):
        self.h2d(self.accum, np.zeros(4 * REF_W * REF_H, np.float32))

    def read_accum(self, rows):
        return self.rd(self.accum, np.float32, 4 * REF_W * rows).reshape(rows, REF_W, 4)

    # -- kernels (argument order = reference src/cl/wavefront.cl) -----------------------------------------
    def generate(self, cam, seeds):
        n = len(seeds)
        assert n % 256 == 0
        d_rays, d_seeds = self.dbuf(nbytes=128 * n), self.dbuf(seeds)
        self.launch("generate", n, 256, [d_rays, self.d_set, d_seeds, np.ascontiguousarray(cam).reshape(1)])
        return self.rd(d_rays, W.Ray, n), self.rd(d_seeds, np.uint32, n)

    def extend(self, rays):
        """One work-group of 256 work-items (the kernel's counter swap sits behind a work-group barrier)."""
        n = len(rays)
        d_rays = self.dbuf(rays)
        self.set_counts(0, n, 0)
        self.launch("extend", 256, 256, [d_rays, self.prims, self.tlas, self.blas, self.nodes, self.idx, self.accum, self.d_set])
        return self.rd(d_rays, W.Ray, n)

    def shade_s0(self, rays, seeds):
        """Global size 1: schedule S0 (descending slots, stream seeds[0])."""
        n = len(rays)
        d_in, d_out, d_sh, d_seeds = self.dbuf(rays), self.dbuf(nbytes=128 * n), self.dbuf(nbytes=96 * n), self.dbuf(seeds)
        self.set_counts(0, n, 0)
        self.launch("shade", 1, 1, [d_in, d_out, d_sh, self.prims, self.tex, self.mats, self.lights, self.d_set, self.accum, d_seeds])
        st = self.get_settings()
        nOut, nSh = int(st["numOutRays"]), int(st["shadowRays"])
        return self.rd(d_out, W.Ray, n)[:nOut].copy(), self.rd(d_sh, W.ShadowRay, n)[:nSh].copy(), self.rd(d_seeds, np.uint32, len(seeds))

    def connect_s0(self, shadow):
        d_sh = self.dbuf(shadow)
        self.set_counts(0, 0, len(shadow))
        self.launch("connect", 1, 1, [d_sh, self.tlas, self.blas, self.nodes, self.idx, self.prims, self.mats, self.d_set, self.accum])

    def focus(self, x, y, cam):
        self.launch("focus", 1, 1, [np.array([x], np.int32), np.array([y], np.int32), self.tlas, self.blas, self.nodes, self.idx,
                                    self.prims, self.d_set, np.ascontiguousarray(cam).reshape(1)])
        return np.float32(self.get_settings()["focalLength"])

    def close(self):
        for p in self._bufs:
            self.R.ref_free(p)
        self._bufs = []
        self.R.ref_unload(self.mod)
