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

    def extend(self, rays, renderBVH=False):
        """One work-group of 256 work-items (the kernel's counter swap sits behind a work-group barrier).
        renderBVH: also returns the reference's own heat-map values accum[slot] = steps / 255 (wavefront.cl:66-67)."""
        n = len(rays)
        d_rays = self.dbuf(rays)
        self.settings["renderBVH"] = int(renderBVH)
        self.set_counts(0, n, 0)
        self.launch("extend", 256, 256, [d_rays, self.prims, self.tlas, self.blas, self.nodes, self.idx, self.accum, self.d_set])
        self.settings["renderBVH"] = 0
        out = self.rd(d_rays, W.Ray, n)
        if renderBVH:
            return out, self.rd(self.accum, np.float32, 4 * n).reshape(n, 4)
        return out

    def shade_s0(self, rays, seeds):
        """Global size 1: schedule S0 (descending slots, stream seeds[0])."""
        n = len(rays)
        d_in, d_out, d_sh, d_seeds = self.dbuf(rays), self.dbuf(nbytes=128 * n), self.dbuf(nbytes=96 * n), self.dbuf(seeds)
        self.set_counts(0, n, 0)
        self.launch("shade", 1, 1, [d_in, d_out, d_sh, self.prims, self.tex, self.mats, self.lights, self.d_set, self.accum, d_seeds])
        st = self.get_settings()
        nOut, nSh = int(st["numOutRays"]), int(st["shadowRays"])
        return self.rd(d_out, W.Ray, n)[:nOut].copy(), self.rd(d_sh, W.ShadowRay, n)[:nSh].copy(), self.rd(d_seeds, np.uint32, len(seeds))

    def shade_s1(self, rays, seeds):
        """The reference's shade kernel under schedule S1 (queue slot g is shaded with RNG stream seeds[g], survivors keep their
        order): one launch of ONE work-item per ray, in slot order, with the ray / seed pointers advanced to the slot and the
        extension-ray pointer to the number appended so far (the kernel resets numOutRays itself, wavefront.cl:90-93; the shadow
        counter keeps running).  A legal execution of the reference - and the schedule the HIP path implements."""
        n = len(rays)
        d_in, d_out, d_sh, d_seeds = self.dbuf(rays), self.dbuf(nbytes=128 * max(n, 1)), self.dbuf(nbytes=96 * max(n, 1)), self.dbuf(seeds)
        n_out = n_sh = 0
        for slot in range(n):
            self.set_counts(0, 1, n_sh)
            self.launch("shade", 1, 1, [C.c_void_p(d_in.value + 128 * slot), C.c_void_p(d_out.value + 128 * n_out), d_sh, self.prims, self.tex,
                                        self.mats, self.lights, self.d_set, self.accum, C.c_void_p(d_seeds.value + 4 * slot)])
            st = self.get_settings()
            n_out += int(st["numOutRays"])
            n_sh = int(st["shadowRays"])
        return (self.rd(d_out, W.Ray, max(n, 1))[:n_out].copy(), self.rd(d_sh, W.ShadowRay, max(n, 1))[:n_sh].copy(),
                self.rd(d_seeds, np.uint32, len(seeds)))

    def frame_s1(self, cam, y0, y1, shading=1, russian_roulette=True, bounces=7):
        """One whole Renderer::RayTrace() (renderer.cpp:64-94) over rows [y0, y1) through the reference's own kernels under schedule S1:
        generate; 7 x { extend (one work-group), shade (one single-work-item launch per queue slot, RNG stream seeds[slot], survivors
        appended in slot order), [connect if !RR and NEE], swap }; [connect if RR] - connect likewise one launch per shadow ray in
        queue order, so a pixel's contributions are added in (bounce, slot) order.  This is the schedule the HIP path implements:
        its accumulator and per-slot RNG states can be compared with this frame directly."""
        first, n, n_all = y0 * REF_W, (y1 - y0) * REF_W, y1 * REF_W
        assert n_all % 256 == 0
        from oracle.oracle_py import seed_stream
        d_all, d_seeds_all = self.dbuf(nbytes=128 * n_all), self.dbuf(seed_stream(0, n_all))
        self.set_counts(0, n, 0)
        self.launch("generate", n_all, 256, [d_all, self.d_set, d_seeds_all, np.ascontiguousarray(cam).reshape(1)])
        ray1, ray2 = C.c_void_p(d_all.value + 128 * first), self.dbuf(nbytes=128 * n)
        d_seeds = C.c_void_p(d_seeds_all.value + 4 * first)
        d_sh = self.dbuf(nbytes=96 * n * bounces)
        self.clear_accum()
        nee = shading == 1
        counts, n_in, n_sh = [], n, 0

        def connect_all(n_shadow):
            for k in range(n_shadow):
                self.set_counts(0, 0, 1)
                self.launch("connect", 1, 1, [C.c_void_p(d_sh.value + 96 * k), self.tlas, self.blas, self.nodes, self.idx, self.prims, self.mats, self.d_set, self.accum])

        for b in range(bounces):
            self.set_counts(0, n_in, n_sh)
            self.launch("extend", 256, 256, [ray1, self.prims, self.tlas, self.blas, self.nodes, self.idx, self.accum, self.d_set])
            if not russian_roulette:
                n_sh = 0                                               # extend resets the counter (wavefront.cl:54-56)
            counts.append(n_in)
            n_out = 0
            for slot in range(n_in):
                self.set_counts(0, 1, n_sh)
                self.launch("shade", 1, 1, [C.c_void_p(ray1.value + 128 * slot), C.c_void_p(ray2.value + 128 * n_out), d_sh, self.prims, self.tex,
                                            self.mats, self.lights, self.d_set, self.accum, C.c_void_p(d_seeds.value + 4 * slot)])
                st = self.get_settings()
                n_out += int(st["numOutRays"])
                n_sh = int(st["shadowRays"])
            if not russian_roulette and nee:
                connect_all(n_sh)
            ray1, ray2 = ray2, ray1
            n_in = n_out
        if russian_roulette and nee:
            connect_all(n_sh)
        return dict(n_in=counts, n_left=n_in, n_shadow=n_sh, seeds=self.rd(d_seeds, np.uint32, n),
                    accum=self.rd(self.accum, np.float32, 4 * REF_W * y1).reshape(y1 * REF_W, 4)[first:].copy())

    def connect_s0(self, shadow):
        d_sh = self.dbuf(shadow)
        self.set_counts(0, 0, len(shadow))
        self.launch("connect", 1, 1, [d_sh, self.tlas, self.blas, self.nodes, self.idx, self.prims, self.mats, self.d_set, self.accum])

    def frame_s0(self, cam, y0, y1, shading=1, russian_roulette=True, seeds_first=None, bounces=7):
        """One whole Renderer::RayTrace() (reference src/renderer.cpp:64-94) over the rows [y0, y1) of the reference's
        compile-time 1280x720 frame, driven through the reference's own kernels in the reference's order:
            settings{numInRays 0, numOutRays n, shadowRays 0};  generate;
            7 x { extend (one work-group of 256), shade (ONE work-item = schedule S0), [connect if !RR and NEE], swap };
            [connect if RR].
        generate derives the pixel from get_global_id (wavefront.cl:26-33), so it is launched over rows [0, y1) and the band's
        rays / seeds are the tail of what it wrote; shade's RNG stream is the band's seeds[0] (wavefront.cl:97 with one
        work-item).  Returns the per-bounce captures (rays after extend, queue lengths, RNG state) and the accumulator."""
        first, n, n_all = y0 * REF_W, (y1 - y0) * REF_W, y1 * REF_W
        assert n_all % 256 == 0
        from oracle.oracle_py import seed_stream
        seeds_all = seed_stream(0, n_all)
        d_all, d_seeds_all = self.dbuf(nbytes=128 * n_all), self.dbuf(seeds_all)
        self.set_counts(0, n, 0)                                      # renderer.cpp:66-69
        self.launch("generate", n_all, 256, [d_all, self.d_set, d_seeds_all, np.ascontiguousarray(cam).reshape(1)])
        gen = self.rd(d_all, W.Ray, n_all)[first:].copy()
        seeds = self.rd(d_seeds_all, np.uint32, n_all)[first:].copy()
        ray1, ray2 = self.dbuf(gen), self.dbuf(nbytes=128 * n)
        d_sh, d_seeds = self.dbuf(nbytes=96 * n * bounces), self.dbuf(seeds)
        self.clear_accum()
        cap = dict(gen=gen, gen_seeds=seeds.copy(), ext=[], n_in=[], n_out=[], n_shadow=[], seed0=[], shadow=None)
        nee = shading == 1
        n_in = n
        for b in range(bounces):                                      # renderer.cpp:75-90
            self.launch("extend", 256, 256, [ray1, self.prims, self.tlas, self.blas, self.nodes, self.idx, self.accum, self.d_set])
            cap["n_in"].append(n_in)
            cap["ext"].append(self.rd(ray1, W.Ray, n)[:n_in].copy())
            self.launch("shade", 1, 1, [ray1, ray2, d_sh, self.prims, self.tex, self.mats, self.lights, self.d_set, self.accum, d_seeds])
            st = self.get_settings()
            n_in = int(st["numOutRays"])
            cap["n_out"].append(n_in)
            cap["n_shadow"].append(int(st["shadowRays"]))
            cap["seed0"].append(int(self.rd(d_seeds, np.uint32, 1)[0]))
            if not russian_roulette and nee:                          # renderer.cpp:85-87
                if cap["shadow"] is None:
                    cap["shadow"] = []
                cap["shadow"].append(self.rd(d_sh, W.ShadowRay, max(int(st["shadowRays"]), 1))[:int(st["shadowRays"])].copy())
                self.launch("connect", 1, 1, [d_sh, self.tlas, self.blas, self.nodes, self.idx, self.prims, self.mats, self.d_set, self.accum])
            ray1, ray2 = ray2, ray1
        cap["last_out"] = self.rd(ray1, W.Ray, n)[:n_in].copy()       # appended by the 7th shade, never traced
        if russian_roulette:                                          # renderer.cpp:91-92
            ns = int(self.get_settings()["shadowRays"])
            cap["shadow"] = [self.rd(d_sh, W.ShadowRay, max(ns, 1))[:ns].copy()]
            self.launch("connect", 1, 1, [d_sh, self.tlas, self.blas, self.nodes, self.idx, self.prims, self.mats, self.d_set, self.accum])
        cap["accum"] = self.rd(self.accum, np.float32, 4 * REF_W * y1).reshape(y1 * REF_W, 4)[first:].copy()
        cap["first_pixel"] = first
        return cap

    def focus(self, x, y, cam):
        self.launch("focus", 1, 1, [np.array([x], np.int32), np.array([y], np.int32), self.tlas, self.blas, self.nodes, self.idx,
                                    self.prims, self.d_set, np.ascontiguousarray(cam).reshape(1)])
        return np.float32(self.get_settings()["focalLength"])

    def close(self):
        for p in self._bufs:
            self.R.ref_free(p)
        self._bufs = []
        self.R.ref_unload(self.mod)


class RefPost:
    """The reference's post-processing kernels (src/cl/postproc.cl compiled unmodified into oracle/_ref/postproc.co): prep ->
    [vignetting] -> [gammaCorr] -> [chromatic] over two swap buffers, in the order and under the conditions of Renderer::PostProc
    (src/renderer.cpp:95-124).  The kernels take the pixel from get_global_id at the compile-time 1280x720, so they are launched
    over the first `rows` rows.  `display` / `saveImage` go through a GL image and are not run; what is returned is the swap
    buffer `display` would show (float3 with 16-byte stride)."""

    def __init__(self):
        self.R = C.CDLL(os.path.join(REF_DIR, "libref_runner.so"))
        self.R.ref_last_error.restype = C.c_char_p
        self._chk(self.R.ref_init(0))
        self.mod = C.c_void_p()
        self._chk(self.R.ref_load(os.path.join(REF_DIR, "postproc.co").encode(), C.byref(self.mod)))

    _chk = RefGPU._chk

    def run(self, accum_rows, frames, vignette, gamma, chromatic):
        a = np.ascontiguousarray(accum_rows, dtype=np.float32).reshape(-1, 4)
        n = len(a)
        assert n % 256 == 0
        bufs = []

        def dbuf(arr=None, nbytes=None):
            p = C.c_void_p()
            nb = arr.nbytes if arr is not None else nbytes
            self._chk(self.R.ref_malloc(C.byref(p), C.c_size_t(nb)))
            if arr is not None:
                self._chk(self.R.ref_h2d(p, arr.ctypes.data_as(C.c_void_p), C.c_size_t(nb)))
            bufs.append(p)
            return p

        def launch(name, args):
            hold = [x if isinstance(x, np.ndarray) else C.c_void_p(x.value) for x in args]
            arr = (C.c_void_p * len(args))(*[h.ctypes.data_as(C.c_void_p) if isinstance(h, np.ndarray) else C.cast(C.pointer(h), C.c_void_p) for h in hold])
            self._chk(self.R.ref_launch(self.mod, name.encode(), n, 256, arr))

        st = np.zeros(1, dtype=W.Settings)
        st["frames"] = frames
        d_acc, src, dst, d_set = dbuf(a), dbuf(nbytes=16 * n), dbuf(nbytes=16 * n), dbuf(st)
        launch("prep", [d_acc, src, d_set])                                   # renderer.cpp:101-102
        if vignette > 0:                                                      # :103-108
            launch("vignetting", [src, dst, np.array([vignette], np.float32)])
            src, dst = dst, src
        if gamma != 1:                                                        # :109-114
            launch("gammaCorr", [src, dst, np.array([gamma], np.float32)])
            src, dst = dst, src
        if chromatic > 0:                                                     # :115-120
            launch("chromatic", [src, dst, np.array([chromatic], np.float32)])
            src, dst = dst, src
        out = np.zeros((n, 4), np.float32)
        self._chk(self.R.ref_d2h(out.ctypes.data_as(C.c_void_p), src, C.c_size_t(out.nbytes)))
        for p in bufs:
            self.R.ref_free(p)
        return out

    def close(self):
        self.R.ref_unload(self.mod)


def post_available():
    return os.path.exists(os.path.join(REF_DIR, "libref_runner.so")) and os.path.exists(os.path.join(REF_DIR, "postproc.co"))
