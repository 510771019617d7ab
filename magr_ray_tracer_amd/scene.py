"""Scene: Python face of the host-side Scene / BVH2 / BVH4 / TLAS mirror (librt355_host.so).

Method names and argument meaning follow the reference (src/scene.h:5-34, src/bvh.h:4-56,
src/tlas.h:2-12); the heavy lifting (normals, Heron areas, binned-SAH / SBVH build, 4-wide
collapse, TLAS clustering) is C++ (magr_ray_tracer_amd/host/*.cpp).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib


def _view(fn, scene, dtype):
    n = C.c_int(0)
    p = fn(scene, C.byref(n))
    if not p or n.value == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n.value * dtype.itemsize)).from_address(p)
    return np.frombuffer(buf, dtype=dtype, count=n.value).copy()


@dataclass
class SceneArrays:
    """The flat arrays Renderer::InitBuffers uploads (reference src/renderer.cpp:145-208)."""
    prims: np.ndarray
    mats: np.ndarray
    tex: np.ndarray
    lights: np.ndarray
    bvh2: np.ndarray
    bvh4: np.ndarray
    primIdx: np.ndarray
    tlas: np.ndarray
    blas: np.ndarray

    def nodes(self, accel):
        return self.bvh4 if accel == _lib.ACCEL_BVH4 else self.bvh2


def material(color=(0, 0, 0), specular=0.0, n1=0.0, n2=0.0, dielectric=False, absorption=(0, 0, 0), light=False,
             emittance=(0, 0, 0)):
    m = np.zeros((), dtype=_lib.Material)
    m["color"][:3] = color
    m["absorption"][:3] = absorption
    m["specular"], m["n1"], m["n2"] = specular, n1, n2
    m["isDielectric"], m["isLight"] = int(dielectric), int(light)
    m["texIdx"] = -1
    m["emittance"][:3] = emittance
    return m


class Scene:
    def __init__(self):
        self._lib = _lib.host_lib()
        self._h = self._lib.rth_scene_create()
        if not self._h:
            raise RuntimeError("rth_scene_create failed")
        self.num_prims = 0

    def close(self):
        if self._h:
            self._lib.rth_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(self._lib.rth_last_error().decode())
        return rc

    # reference: Scene::AddMaterial (scene.cpp:84-100)
    def AddMaterial(self, name, mat=None):
        m = np.ascontiguousarray(mat) if mat is not None else None
        return self._chk(self._lib.rth_add_material(self._h, name.encode(), _lib.ptr(m) if m is not None else None))

    def AddTexture(self, name, texels):
        t = np.ascontiguousarray(texels, dtype=np.float32)
        h, w = t.shape[:2]
        assert t.shape[2] == 4
        return self._chk(self._lib.rth_add_texture(self._h, name.encode(), _lib.ptr(t), w, h))

    # reference: Scene::LoadTexture (scene.cpp:244-256); PNG, JPEG, TGA and Radiance HDR files
    def LoadTexture(self, filename, name):
        """Read an image file into the texture atlas and add a material `name` that points at it; returns the material index."""
        return self._chk(self._lib.rth_load_texture(self._h, str(filename).encode(), name.encode()))

    def AddSphere(self, pos, radius, material):
        self._chk(self._lib.rth_add_sphere(self._h, _lib.fvec(pos), float(radius), material.encode()))
        self.num_prims += 1

    def AddPlane(self, N, d, material):
        self._chk(self._lib.rth_add_plane(self._h, _lib.fvec(N), float(d), material.encode()))
        self.num_prims += 1

    def AddTriangle(self, v0, v1, v2, material, uv0=(0, 0), uv1=(0, 0), uv2=(0, 0), flipNormal=False):
        self._chk(self._lib.rth_add_triangle(self._h, _lib.fvec(v0), _lib.fvec(v1), _lib.fvec(v2), _lib.fvec(uv0),
                                             _lib.fvec(uv1), _lib.fvec(uv2), material.encode(), int(flipNormal)))
        self.num_prims += 1

    def AddQuad(self, v0, v1, v2, v3, material, flipNormal=False):
        self._chk(self._lib.rth_add_quad(self._h, _lib.fvec(v0), _lib.fvec(v1), _lib.fvec(v2), _lib.fvec(v3),
                                         material.encode(), int(flipNormal)))
        self.num_prims += 2

    def AddTriangles(self, verts, material, uvs=None, flipNormal=False):
        """verts: (n,3,3) float32 triangle soup; uvs: (n,3,2) or None."""
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
        u = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 6)
        self._chk(self._lib.rth_add_triangles(self._h, _lib.ptr(v), _lib.ptr(u) if u is not None else None, v.shape[0],
                                              material.encode(), int(flipNormal)))
        self.num_prims += v.shape[0]

    # reference: Scene::LoadModel (scene.cpp:178-243)
    def LoadModel(self, filename, defaultMaterial, pos=(0, 0, 0), forceDefaultMat=False):
        n = self._chk(self._lib.rth_load_model(self._h, str(filename).encode(), defaultMaterial.encode(), _lib.fvec(pos), int(forceDefaultMat)))
        self.num_prims += n
        return n

    # reference: BVH2::BuildBLAS (bvh.cpp:46-82), bvh2->alpha = 1 -> plain SAH BVH, 0 -> full SBVH
    def BuildBLAS(self, startIdx=0, alpha=1.0, threads=1):
        """threads > 1: task-parallel build, numbered afterwards in the reference's LIFO order (identical arrays)."""
        self._lib.rth_set_build_threads(self._h, int(threads))
        self._chk(self._lib.rth_build_blas(self._h, int(startIdx), float(alpha)))

    def BuildBVH4(self):
        self._chk(self._lib.rth_build_bvh4(self._h))

    def BuildTLAS(self):
        self._chk(self._lib.rth_build_tlas(self._h))

    def SetInstanceTransform(self, blas, invT):
        self._chk(self._lib.rth_set_instance_transform(self._h, int(blas), _lib.fvec(np.asarray(invT, dtype=np.float32).ravel())))

    def stats(self):
        u = np.zeros(5, dtype=np.uint32)
        f = np.zeros(2, dtype=np.float32)
        self._lib.rth_bvh_stats(self._h, _lib.ptr(u), _lib.ptr(f))
        return {"depth": int(u[0]), "nodes": int(u[1]), "spatial_splits": int(u[2]), "prims_clipped": int(u[3]),
                "prims": int(u[4]), "sah_cost": float(f[0]), "build_ms": float(f[1])}

    def texture_array(self):
        """Scene::textures as an (n, 4) float32 view (no acceleration structure needed)."""
        return _view(self._lib.rth_textures, self._h, np.dtype((np.float32, 4)))

    def material_array(self):
        return _view(self._lib.rth_materials, self._h, _lib.Material)

    def arrays(self, bvh4=True):
        L = self._lib
        if bvh4:
            self.BuildBVH4()
        self.BuildTLAS()
        return SceneArrays(
            prims=_view(L.rth_primitives, self._h, _lib.Primitive), mats=_view(L.rth_materials, self._h, _lib.Material),
            tex=_view(L.rth_textures, self._h, np.dtype((np.float32, 4))), lights=_view(L.rth_lights, self._h, np.dtype("<u4")),
            bvh2=_view(L.rth_bvh2_nodes, self._h, _lib.BVHNode2), bvh4=_view(L.rth_bvh4_nodes, self._h, _lib.BVHNode4),
            primIdx=_view(L.rth_prim_idx, self._h, np.dtype("<u4")), tlas=_view(L.rth_tlas_nodes, self._h, _lib.TLASNode),
            blas=_view(L.rth_blas_nodes, self._h, _lib.BVHInstance))


def make_camera(width, height, origin, forward, fov=110.0, aperture=0.1, focalLength=1.0, type=0):
    """CameraManager(fov, type) + UpdateCamVec() (reference src/camera.h:24-34,101-121); camera looks along -forward."""
    cam = np.zeros((), dtype=_lib.Camera)
    rc = _lib.host_lib().rth_camera(int(width), int(height), float(fov), int(type), _lib.fvec(origin), _lib.fvec(forward),
                                    float(aperture), float(focalLength), cam.ctypes.data_as(C.c_void_p))
    if rc < 0:
        raise RuntimeError(_lib.host_lib().rth_last_error().decode())
    return cam


def save_png(path, image):
    """SaveImageF (template/template.cpp:1629-1644): (H,W,4) float image -> 8-bit RGB PNG."""
    a = np.ascontiguousarray(image, dtype=np.float32)
    if _lib.host_lib().rth_save_png(str(path).encode(), a.shape[1], a.shape[0], _lib.ptr(a)) < 0:
        raise RuntimeError(_lib.host_lib().rth_last_error().decode())
