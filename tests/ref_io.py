"""TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/libref_io.so - the reference's own asset readers (its vendored stb_image,
stb_image_write and tinyobjloader headers compiled where they lie by oracle/build_ref.sh behind oracle/ref_io_runner.cpp)."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_io.so")
_lib = None


def available():
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        L.ref_io_last_error.restype = C.c_char_p
        ip = C.POINTER(C.c_int)
        L.ref_load_image_f.argtypes = [C.c_char_p, ip, ip, ip, C.c_void_p]
        L.ref_load_image_u8.argtypes = [C.c_char_p, ip, ip, ip, C.c_void_p]
        L.ref_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.ref_obj_load.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_int]
        L.ref_obj_faces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for f in (L.ref_obj_tex_name, L.ref_obj_material_name, L.ref_obj_material_diffuse):
            f.restype, f.argtypes = C.c_char_p, [C.c_int]
        _lib = L
    return _lib


def _err():
    return lib().ref_io_last_error().decode()


def load_image_f(path):
    """The reference's LoadImageF (template/template.cpp:1613-1627): (h, w, 3) float32 and stb's channel count."""
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    if lib().ref_load_image_f(str(path).encode(), w, h, c, None) != 0:
        raise RuntimeError(_err())
    out = np.zeros((h.value, w.value, 3), np.float32)
    if lib().ref_load_image_f(str(path).encode(), w, h, c, out.ctypes.data_as(C.c_void_p)) != 0:
        raise RuntimeError(_err())
    return out, c.value


def load_image_u8(path):
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    if lib().ref_load_image_u8(str(path).encode(), w, h, c, None) != 0:
        raise RuntimeError(_err())
    out = np.zeros((h.value, w.value, c.value), np.uint8)
    if lib().ref_load_image_u8(str(path).encode(), w, h, c, out.ctypes.data_as(C.c_void_p)) != 0:
        raise RuntimeError(_err())
    return out


def write_png(path, rgb8):
    a = np.ascontiguousarray(rgb8, np.uint8)
    if lib().ref_write_png(str(path).encode(), a.shape[1], a.shape[0], a.ctypes.data_as(C.c_void_p)) != 0:
        raise RuntimeError("stbi_write_png failed")


def obj_load(path, default_mat="white", pos=(0.0, 0.0, 0.0), force_default=False):
    """Scene::LoadModel (src/scene.cpp:178-243) up to the AddTriangle calls: per triangle the face's `vertices` list after
    std::reverse (+ pos) as (n, 3, 3), its `texcoords` list as (n, 3, 2), the `tex` string as an index into `tex_names`; plus the
    MTL materials (name, diffuse_texname) whose images LoadModel loads first."""
    L = lib()
    n = L.ref_obj_load(str(path).encode(), default_mat.encode(), pos[0], pos[1], pos[2], int(force_default))
    if n < 0:
        raise RuntimeError(_err())
    v, uv, tex = np.zeros((n, 3, 3), np.float32), np.zeros((n, 3, 2), np.float32), np.zeros(n, np.int32)
    L.ref_obj_faces(v.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p), tex.ctypes.data_as(C.c_void_p))
    names = [L.ref_obj_tex_name(i).decode() for i in range(L.ref_obj_tex_count())]
    mtl = [(L.ref_obj_material_name(i).decode(), L.ref_obj_material_diffuse(i).decode()) for i in range(L.ref_obj_material_count())]
    return dict(verts=v, uvs=uv, tex=tex, tex_names=names, materials=mtl)
