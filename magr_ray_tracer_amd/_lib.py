"""ctypes bindings of the two native libraries and numpy dtypes of the wire format.

librt355.so (HIP kernels + C-ABI of include/rt355.h) is mandatory: importing a device entry
point without it raises — there is no Python or CPU fallback for the hot path.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))

# ---- numpy dtypes (include/rt355_types.h; reference src/common.h:3-116) -----------------------
f4 = np.dtype((np.float32, 4))
Ray = np.dtype([("O", f4), ("D", f4), ("rD", f4), ("N", f4), ("I", f4), ("intensity", f4), ("t", "<f4"),
                ("primIdx", "<i4"), ("bounces", "<i4"), ("pixelIdx", "<i4"), ("inside", "u1"), ("lastSpecular", "u1"),
                ("_pad0", "u1", 2), ("u", "<f4"), ("v", "<f4"), ("_pad1", "<u4")], align=False)
ShadowRay = np.dtype([("I", f4), ("L", f4), ("Nl", f4), ("intensity", f4), ("BRDF", f4), ("lightIdx", "<i4"),
                      ("pixelIdx", "<i4"), ("dotNL", "<f4"), ("dist", "<f4")])
Material = np.dtype([("color", f4), ("absorption", f4), ("specular", "<f4"), ("n1", "<f4"), ("n2", "<f4"),
                     ("isDielectric", "u1"), ("_pad0", "u1", 3), ("texIdx", "<i4"), ("texW", "<i4"), ("texH", "<i4"),
                     ("isLight", "u1"), ("_pad1", "u1", 3), ("emittance", f4)])
Primitive = np.dtype([("v0", f4), ("v1", f4), ("v2", f4), ("N", f4), ("centroid", f4), ("uv0", "<f4", 2), ("uv1", "<f4", 2),
                      ("uv2", "<f4", 2), ("_padt", "<f4", 2), ("objType", "<i4"), ("matIdx", "<i4"), ("area", "<f4"),
                      ("_pad", "<u4")])  # triangle view of the 128-byte union; spheres/planes overlay v0/v1
Camera = np.dtype([("type", "<i4"), ("fov", "<f4"), ("aperture", "<f4"), ("focalLength", "<f4"), ("forward", f4),
                   ("right", f4), ("up", f4), ("origin", f4), ("horizontal", f4), ("vertical", f4), ("topLeft", f4)])
Settings = np.dtype([("numPrimitives", "<i4"), ("numLights", "<i4"), ("tracerType", "<i4"), ("frames", "<i4"),
                     ("antiAliasing", "<i4"), ("numInRays", "<i4"), ("numOutRays", "<i4"), ("shadowRays", "<i4"),
                     ("renderBVH", "<i4"), ("focalLength", "<f4")])
BVHNode2 = np.dtype([("aabbMin", f4), ("aabbMax", f4), ("first", "<u4"), ("count", "<u4"), ("_pad", "<u4", 2)])
BVHNode4 = np.dtype([("aabbMin", f4, 4), ("aabbMax", f4, 4), ("first", "<i4", 4), ("count", "<i4", 4)])
BVHInstance = np.dtype([("bvhIdx", "<u4"), ("invT", "<f4", 16)])
TLASNode = np.dtype([("aabbMin", f4), ("aabbMax", f4), ("leftRight", "<u4"), ("BLASidx", "<u4"), ("_pad", "<u4", 2)])
ShadowRecord = np.dtype([("o", "<f4", 3), ("tmax", "<f4"), ("l", "<f4", 3), ("pixelIdx", "<i4"), ("radiance", f4)])
Counters = np.dtype([(n, "<u8") for n in (
    "extend_rays", "extend_tlas_visits", "extend_inst_visits", "extend_node_visits", "extend_prim_tests",
    "connect_rays", "connect_tlas_visits", "connect_inst_visits", "connect_node_visits", "connect_prim_tests",
    "primary_rays", "shadow_rays", "frames", "extend_node_issues", "extend_leaf_issues", "connect_node_issues", "connect_leaf_issues",
    "extend_loop_node_events", "extend_loop_leaf_events", "connect_loop_node_events", "connect_loop_leaf_events")])
StageTimes = np.dtype([(n, "<f8") for n in ("generate_ms", "extend_ms", "shade_ms", "compact_ms", "connect_ms", "accumulate_ms")] +
                      [(n, "<i8") for n in ("generate_launches", "extend_launches", "shade_launches", "compact_launches",
                                            "connect_launches", "accumulate_launches")])
Config = np.dtype([(n, "<i4") for n in ("width", "height", "y0", "y1", "max_bounces", "shading", "sampling", "accel",
                                         "russian_roulette", "filter_fireflies", "device", "extend_variant", "profile",
                                         "shade_blocks_per_cu", "persist_blocks_per_cu")] +
                  [("reserved", "<i4", 1)])

KernelInfo = np.dtype([(n, "<i4") for n in ("layout", "persist", "persist4", "stack_entries", "persist_grid", "persist_grid_connect",
                                             "shade_grid", "n_blas")])

_SIZES = {"Ray": (Ray, 128), "ShadowRay": (ShadowRay, 96), "Material": (Material, 80), "Primitive": (Primitive, 128),
          "Camera": (Camera, 128), "Settings": (Settings, 40), "BVHNode2": (BVHNode2, 48), "BVHNode4": (BVHNode4, 160),
          "BVHInstance": (BVHInstance, 68), "TLASNode": (TLASNode, 48), "ShadowRecord": (ShadowRecord, 48), "Config": (Config, 64)}
for _n, (_d, _s) in _SIZES.items():
    assert _d.itemsize == _s, (_n, _d.itemsize, _s)

SHADING_SIMPLE, SHADING_NEE = 0, 1
SAMPLING_HEMISPHERE, SAMPLING_COSINE = 0, 1
ACCEL_BVH2, ACCEL_BVH4 = 0, 1
PRIM_SPHERE, PRIM_PLANE, PRIM_TRIANGLE = 0, 1, 2
MAX_BOUNCES = 7

DEVICE_SYMBOLS = [
    "rt_last_error", "rt_device_count", "rt_kernel_info", "rt_create", "rt_destroy", "rt_upload_scene", "rt_share_scene", "rt_set_seeds", "rt_seed_default",
    "rt_get_seeds", "rt_bind_accum", "rt_accum_device_ptr", "rt_stream", "rt_reset", "rt_render", "rt_synchronize", "rt_focus",
    "rt_read_accum", "rt_write_accum", "rt_postproc", "rt_read_counters", "rt_reset_counters", "rt_read_stage_times", "rt_reset_stage_times", "rt_set_profile",
    "rt_stage_begin_frame", "rt_stage_generate", "rt_stage_extend", "rt_stage_shade", "rt_stage_connect",
    "rt_debug_get_rays", "rt_debug_set_rays", "rt_debug_get_shadow", "rt_debug_enable_steps", "rt_debug_get_steps", "rt_validate_scene",
    "rt_group_create", "rt_group_destroy", "rt_group_lanes", "rt_group_concurrency", "rt_group_lane", "rt_group_frames", "rt_group_upload_scene", "rt_group_share_scene",
    "rt_group_seed", "rt_group_reset", "rt_group_render", "rt_group_synchronize", "rt_group_sum", "rt_group_read_accum", "rt_group_focus",
    "rt_group_postproc"]
HOST_SYMBOLS = [
    "rth_last_error", "rth_scene_create", "rth_scene_destroy", "rth_add_material", "rth_add_texture", "rth_load_texture", "rth_add_sphere",
    "rth_add_plane", "rth_add_triangle", "rth_add_quad", "rth_add_triangles", "rth_build_blas", "rth_build_bvh4",
    "rth_build_tlas", "rth_bvh4_from_nodes", "rth_set_instance_transform", "rth_primitives", "rth_materials", "rth_textures", "rth_lights",
    "rth_bvh2_nodes", "rth_bvh4_nodes", "rth_prim_idx", "rth_tlas_nodes", "rth_blas_nodes", "rth_bvh_stats", "rth_camera",
    "rth_renderer_create", "rth_renderer_destroy", "rth_renderer_init", "rth_renderer_set_camera", "rth_renderer_tick",
    "rth_renderer_read", "rth_renderer_camera", "rth_seed_stream", "rth_load_model", "rth_save_png",
    "rth_set_build_threads", "rth_renderer_save_frame", "rth_renderer_camera_move", "rth_renderer_camera_mouse", "rth_renderer_camera_zoom", "rth_renderer_frames", "rth_renderer_set_lanes"]

_dev = None
_host = None


class NativeLibraryMissing(RuntimeError):
    pass


def _load(name, local=False):
    path = os.path.join(_PKG, name)
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            f"{path} is missing: build it with `python -m magr_ray_tracer_amd.build` (hipcc, gfx950). "
            "The hot path has no Python/CPU fallback.")
    return C.CDLL(path, mode=C.RTLD_LOCAL if local else C.RTLD_GLOBAL)   # a variant build exports the same symbols: keep them out of the global namespace


_variants = {}


def device_lib(variant=None):
    """librt355.so with argtypes set. Raises NativeLibraryMissing if it was not built.  variant="refb": librt355_refb.so, the build with
    the reference's OpenCL builtin sequences (-DRT355_REF_BUILTINS; tests against the reference's kernels only)."""
    global _dev
    if variant:
        if variant not in _variants:
            keep, _dev = _dev, None
            try:
                _variants[variant] = _bind_device(_load(f"librt355_{variant}.so", local=True))
            finally:
                _dev = keep
        return _variants[variant]
    if _dev is None:
        _dev = _bind_device(_load("librt355.so"))
    return _dev


def _bind_device(lib):
    if True:
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_create.argtypes = [vp, C.POINTER(vp)]
        lib.rt_kernel_info.argtypes = [vp, vp]
        lib.rt_destroy.argtypes = [vp]
        lib.rt_upload_scene.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32]
        lib.rt_share_scene.argtypes = [vp, vp]
        lib.rt_set_seeds.argtypes = [vp, vp, i64]
        lib.rt_seed_default.argtypes = [vp]
        lib.rt_get_seeds.argtypes = [vp, vp, i64]
        lib.rt_bind_accum.argtypes = [vp, vp]
        lib.rt_accum_device_ptr.argtypes = [vp]
        lib.rt_accum_device_ptr.restype = vp
        lib.rt_stream.argtypes = [vp]
        lib.rt_stream.restype = vp
        lib.rt_reset.argtypes = [vp]
        lib.rt_render.argtypes = [vp, vp, vp, i32]
        lib.rt_synchronize.argtypes = [vp]
        lib.rt_focus.argtypes = [vp, i32, i32, vp, C.POINTER(C.c_float)]
        lib.rt_read_accum.argtypes = [vp, vp]
        lib.rt_read_counters.argtypes = [vp, vp]
        lib.rt_write_accum.argtypes = [vp, vp]
        lib.rt_postproc.argtypes = [vp, i32, C.c_float, C.c_float, C.c_float, vp, vp]
        lib.rt_reset_counters.argtypes = [vp]
        lib.rt_read_stage_times.argtypes = [vp, vp]
        lib.rt_reset_stage_times.argtypes = [vp]
        lib.rt_set_profile.argtypes = [vp, i32]
        lib.rt_stage_begin_frame.argtypes = [vp]
        lib.rt_stage_generate.argtypes = [vp, vp, vp]
        lib.rt_stage_extend.argtypes = [vp, i32, i32]
        lib.rt_stage_shade.argtypes = [vp, i32]
        lib.rt_stage_connect.argtypes = [vp, i32, i32]
        lib.rt_debug_get_rays.argtypes = [vp, i32, vp, i32, C.POINTER(i32)]
        lib.rt_debug_set_rays.argtypes = [vp, i32, vp, i32]
        lib.rt_debug_get_shadow.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
        lib.rt_debug_get_steps.argtypes = [vp, vp, i32, C.POINTER(i32)]
        lib.rt_debug_enable_steps.argtypes = [vp, i32]
        lib.rt_validate_scene.argtypes = [i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32]
        lib.rt_group_create.argtypes = [vp, i32, C.POINTER(vp)]
        lib.rt_group_destroy.argtypes = [vp]
        lib.rt_group_lanes.argtypes = [vp]
        lib.rt_group_concurrency.argtypes = [vp]
        lib.rt_group_lane.argtypes = [vp, i32]
        lib.rt_group_lane.restype = vp
        lib.rt_group_frames.argtypes = [vp]
        lib.rt_group_frames.restype = C.c_uint64
        lib.rt_group_upload_scene.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32]
        lib.rt_group_seed.argtypes = [vp, C.c_uint64]
        lib.rt_group_share_scene.argtypes = [vp, vp]
        lib.rt_group_reset.argtypes = [vp]
        lib.rt_group_render.argtypes = [vp, vp, vp, i32]
        lib.rt_group_synchronize.argtypes = [vp]
        lib.rt_group_sum.argtypes = [vp, vp]
        lib.rt_group_read_accum.argtypes = [vp, vp]
        lib.rt_group_focus.argtypes = [vp, i32, i32, vp, C.POINTER(C.c_float)]
        lib.rt_group_postproc.argtypes = [vp, i32, C.c_float, C.c_float, C.c_float, vp, vp]
    return lib


def host_lib():
    """librt355_host.so (depends on librt355.so) with argtypes set."""
    global _host
    if _host is None:
        device_lib()
        lib = _load("librt355_host.so")
        vp, i32, fp, cp = C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_char_p
        lib.rth_last_error.restype = cp
        lib.rth_scene_create.restype = vp
        lib.rth_scene_destroy.argtypes = [vp]
        lib.rth_add_material.argtypes = [vp, cp, vp]
        lib.rth_add_texture.argtypes = [vp, cp, vp, i32, i32]
        lib.rth_load_texture.argtypes = [vp, cp, cp]
        lib.rth_add_sphere.argtypes = [vp, fp, C.c_float, cp]
        lib.rth_add_plane.argtypes = [vp, fp, C.c_float, cp]
        lib.rth_add_triangle.argtypes = [vp, fp, fp, fp, fp, fp, fp, cp, i32]
        lib.rth_add_quad.argtypes = [vp, fp, fp, fp, fp, cp, i32]
        lib.rth_add_triangles.argtypes = [vp, vp, vp, i32, cp, i32]
        lib.rth_build_blas.argtypes = [vp, i32, C.c_float]
        lib.rth_build_bvh4.argtypes = [vp]
        lib.rth_set_build_threads.argtypes = [vp, i32]
        lib.rth_build_tlas.argtypes = [vp]
        lib.rth_bvh4_from_nodes.argtypes = [vp, i32, vp]
        lib.rth_set_instance_transform.argtypes = [vp, i32, fp]
        for n in ("rth_primitives", "rth_materials", "rth_textures", "rth_lights", "rth_bvh2_nodes", "rth_bvh4_nodes",
                  "rth_prim_idx", "rth_tlas_nodes", "rth_blas_nodes"):
            getattr(lib, n).argtypes = [vp, C.POINTER(i32)]
            getattr(lib, n).restype = vp
        lib.rth_bvh_stats.argtypes = [vp, vp, vp]
        lib.rth_camera.argtypes = [i32, i32, C.c_float, i32, fp, fp, C.c_float, C.c_float, vp]
        lib.rth_renderer_create.argtypes = [vp] + [i32] * 10
        lib.rth_renderer_create.restype = vp
        lib.rth_renderer_destroy.argtypes = [vp]
        lib.rth_renderer_init.argtypes = [vp]
        lib.rth_renderer_set_camera.argtypes = [vp, fp, fp, C.c_float, C.c_float]
        lib.rth_renderer_tick.argtypes = [vp, i32]
        lib.rth_renderer_read.argtypes = [vp, vp, fp]
        lib.rth_renderer_camera.argtypes = [vp, vp]
        lib.rth_seed_stream.argtypes = [vp, C.c_int64, C.c_int64]
        lib.rth_load_model.argtypes = [vp, cp, cp, fp, i32]
        lib.rth_save_png.argtypes = [cp, i32, i32, vp]
        lib.rth_renderer_save_frame.argtypes = [vp, cp]
        lib.rth_renderer_camera_move.argtypes = [vp, i32]
        lib.rth_renderer_camera_mouse.argtypes = [vp, C.c_float, C.c_float]
        lib.rth_renderer_camera_zoom.argtypes = [vp, C.c_float]
        lib.rth_renderer_frames.argtypes = [vp]
        lib.rth_renderer_set_lanes.argtypes = [vp, i32]
        _host = lib
    return _host


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def fvec(v):
    a = (C.c_float * len(v))(*[float(x) for x in v])
    return C.cast(a, C.POINTER(C.c_float))
