"""Procedural stand-ins for the reference's scenes (its .obj assets are git-LFS pointers and
sponza.obj is absent: SURVEY.md Appendix E).  Geometry is generated RNG-free with numpy and fed
through the Scene API (AddMaterial / AddTriangles / AddQuad / AddSphere / BuildBLAS), materials
follow reference src/scene.cpp:14-43.

    cube_scene()      config 1: unit cube (12 tris) on a floor + quad light
    bunny_class()     config 2: closed displaced-sphere mesh, 2*n*n triangles (n=187 -> 69,938)
    sponza_class()    config 3/4: atrium with two arcaded storeys, columns, arches, drapes (~260k tris)
    model_scene()     an OBJ file (a real sponza.obj when supplied) + the reference's light quad
    mixed_scene()     small scene with spheres, a mirror, glass, a texture and two lights (edge cases)
    two_blas_scene()  two BLAS under a TLAS (config 5 shape)
"""
import numpy as np

from .scene import Scene, material, make_camera


def _std_materials(s):
    s.AddMaterial("grey", material(color=(0.231, 0.266, 0.294)))
    s.AddMaterial("white", material(color=(0.8, 0.8, 0.8)))
    s.AddMaterial("red", material(color=(0.75, 0.2, 0.15)))
    s.AddMaterial("green", material(color=(0.2, 0.7, 0.25)))
    s.AddMaterial("sand", material(color=(0.72, 0.62, 0.45)))
    s.AddMaterial("mirror", material(color=(0.1, 0.1, 0.9), specular=0.5))
    s.AddMaterial("white-glass", material(color=(1, 1, 1), dielectric=True, n1=1.0, n2=1.1, specular=0.03, absorption=(0.01,) * 3))
    s.AddMaterial("white-light", material(color=(1.0, 0.7, 0.1), light=True, emittance=(90, 90, 90)))
    s.AddMaterial("green-light", material(color=(0.1, 1.0, 0.1), light=True, emittance=(1, 10, 1)))
    s.AddMaterial("red-light", material(color=(1.0, 0.1, 0.1), light=True, emittance=(100, 10, 10)))


def grid_tris(P):
    """P: (nu+1, nv+1, 3) grid of points -> (2*nu*nv, 3, 3) triangles, winding (u x v)."""
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.stack([a, b, c], axis=-2)
    t2 = np.stack([c, d, a], axis=-2)
    return np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], axis=0).astype(np.float32)


def param_surface(fn, nu, nv, u0=0.0, u1=1.0, v0=0.0, v1=1.0):
    u = np.linspace(u0, u1, nu + 1, dtype=np.float64)
    v = np.linspace(v0, v1, nv + 1, dtype=np.float64)
    U, V = np.meshgrid(u, v, indexing="ij")
    return grid_tris(np.stack(fn(U, V), axis=-1))


def box_tris(lo, hi):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    c = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]],
                 dtype=np.float32)
    q = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]
    t = []
    for a, b, cc, d in q:
        t.append([c[a], c[b], c[cc]])
        t.append([c[cc], c[d], c[a]])
    return np.array(t, dtype=np.float32)


# ----------------------------------------------------------------------------------------- config 1
def cube_scene():
    s = Scene()
    _std_materials(s)
    s.AddTriangles(box_tris((-0.5, 0.0, -0.5), (0.5, 1.0, 0.5)), "red")
    s.AddQuad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), "grey")
    s.AddQuad((-1, 3, -1), (1, 3, -1), (1, 3, 1), (-1, 3, 1), "white-light")  # normal (0,-1,0)
    s.BuildBLAS(0, 1.0)
    view = dict(origin=(1.8, 1.6, 2.6), forward=(0.5, 0.28, 0.82), fov=60.0, aperture=0.0)
    return s, view


# ----------------------------------------------------------------------------------------- config 2
def bunny_class(n=187, alpha=1.0):
    """Closed 'bunny-class' blob: lat-long displaced sphere with 2*n*n triangles, floor, 4x4 quad light."""
    s = Scene()
    _std_materials(s)

    def blob(U, V):
        th, ph = U * 2 * np.pi, V * np.pi
        r = 1.0 + 0.18 * np.sin(5 * th) * np.sin(3 * ph) ** 2 + 0.10 * np.cos(9 * ph + 2 * th) * np.sin(ph) + 0.25 * np.exp(-8 * (ph - 0.6) ** 2) * (1 + np.cos(2 * th))
        x = r * np.sin(ph) * np.cos(th)
        y = 1.25 + 1.1 * r * np.cos(ph)
        z = 0.9 * r * np.sin(ph) * np.sin(th)
        return x, y, z

    s.AddTriangles(param_surface(blob, n, n), "sand")
    s.AddQuad((-10, 0, -10), (-10, 0, 10), (10, 0, 10), (10, 0, -10), "grey")
    s.AddQuad((-2, 6, -2), (2, 6, -2), (2, 6, 2), (-2, 6, 2), "white-light")
    s.BuildBLAS(0, alpha)
    view = dict(origin=(0.6, 2.0, 4.2), forward=(0.12, 0.18, 0.97), fov=60.0, aperture=0.05)
    return s, view


# ----------------------------------------------------------------------------------------- config 3/4
def sponza_class(detail=1.0, alpha=1.0):
    """Atrium ('sponza-class'): 36 x 14 m floor, 14 m high, open roof slot with sky, two storeys of
    fluted columns carrying round arches along both long sides, ribbed walls, six hanging drapes.
    detail=1.0 gives ~262k triangles (Crytek Sponza has 262,267); the generator is RNG-free."""
    s = Scene()
    _std_materials(s)
    d = float(detail)
    k = lambda v: max(2, int(round(v * d * 1.25)))  # noqa: E731
    LX, LZ, H = 18.0, 7.0, 14.0

    # floor (gently cambered paving) and ribbed long walls, end walls, roof with a central slot
    s.AddTriangles(param_surface(lambda U, V: ((U - .5) * 2 * LX, 0.02 * np.sin(U * 90) * np.sin(V * 40), (V - .5) * 2 * LZ), k(96), k(40)), "sand")
    for sign in (-1.0, 1.0):
        wall = param_surface(lambda U, V: ((U - .5) * 2 * LX, V * H, sign * (LZ + 0.08 * np.cos(U * 140))), k(160), k(24))
        s.AddTriangles(wall if sign > 0 else wall[:, ::-1], "white")
        end = param_surface(lambda U, V: (sign * LX + 0 * U, V * H, (U - .5) * 2 * LZ), k(40), k(24))
        s.AddTriangles(end if sign < 0 else end[:, ::-1], "white")
        roof = param_surface(lambda U, V: ((U - .5) * 2 * LX, H + 0 * U, sign * (2.2 + V * (LZ - 2.2))), k(64), k(8))
        s.AddTriangles(roof if sign < 0 else roof[:, ::-1], "grey")

    # colonnades: 2 storeys x 2 sides x 9 columns, fluted shafts with base and capital blocks
    ncol = 9
    xs = np.linspace(-LX + 2.5, LX - 2.5, ncol)
    zc, rcol = 4.4, 0.42
    for storey, (y0, y1) in enumerate(((0.0, 5.6), (6.4, 11.6))):
        for sign in (-1.0, 1.0):
            for cx in xs:
                def shaft(U, V, cx=cx, sign=sign, y0=y0, y1=y1):
                    th = U * 2 * np.pi
                    r = rcol * (1.0 - 0.12 * V) * (1.0 + 0.05 * np.cos(16 * th))
                    return cx + r * np.cos(th), y0 + 0.35 + V * (y1 - y0 - 0.8), sign * zc + r * np.sin(th)
                s.AddTriangles(param_surface(shaft, k(48), k(22)), "sand")
                s.AddTriangles(box_tris((cx - .6, y0, sign * zc - .6), (cx + .6, y0 + .35, sign * zc + .6)), "white")
                s.AddTriangles(box_tris((cx - .62, y1 - .45, sign * zc - .62), (cx + .62, y1, sign * zc + .62)), "white")
            # arches between neighbouring columns (extruded semicircular bands) and the gallery floor above them
            for a, b in zip(xs[:-1], xs[1:]):
                cxm, rad = 0.5 * (a + b), 0.5 * (b - a) - 0.3

                def arch(U, V, cxm=cxm, rad=rad, sign=sign, y1=y1):
                    ang = U * np.pi
                    return cxm + rad * np.cos(ang), y1 + rad * 0.55 * np.sin(ang) - 0.45 * 0 + 0.0, sign * (zc - 0.45 + 0.9 * V)
                s.AddTriangles(param_surface(arch, k(24), k(6)), "white")
            gal = param_surface(lambda U, V, sign=sign, y1=y1: ((U - .5) * 2 * (LX - 1.0), y1 + 0.8 + 0 * U, sign * (zc - 0.7 + V * (LZ - zc + 0.7))), k(72), k(6))
            s.AddTriangles(gal if sign < 0 else gal[:, ::-1], "grey")

    # drapes: wavy cloth hanging across the nave
    for i, cx in enumerate(np.linspace(-12.0, 12.0, 6)):
        def drape(U, V, cx=cx, i=i):
            z = (U - .5) * 5.2
            y = 12.6 - V * 5.5 - 0.5 * np.cos(U * np.pi * 2) * V
            x = cx + 0.28 * np.sin(U * 18 + i) * (0.3 + V) + 0.12 * np.sin(V * 14 + 2 * i)
            return x, y, z
        s.AddTriangles(param_surface(drape, k(64), k(64)), ("red", "green", "sand")[i % 3])

    # one emissive quad as in reference scene.cpp:67 (4 x 4), here under the roof slot facing down
    first_extra = s.num_prims
    s.AddQuad((-2, 13.2, -2), (2, 13.2, -2), (2, 13.2, 2), (-2, 13.2, 2), "white-light")
    del first_extra
    s.BuildBLAS(0, alpha)
    view = dict(origin=(-15.0, 3.2, 0.6), forward=(-0.97, -0.10, -0.05), fov=75.0, aperture=0.02)
    return s, view


def model_scene(path, alpha=1.0, default_material="white"):
    """The reference's model branch (src/scene.cpp:63-69): LoadModel(path, "white") - OBJ + MTL + diffuse textures through the
    tinyobjloader / stb_image-exact readers of the host library - plus one 4 x 4 emissive quad at the reference's coordinates, one
    BLAS over everything, seen from CameraManager's defaults (src/camera.h:24-34).  The hook for a real assets/sponza/sponza.obj
    (absent from the reference checkout): `python bench.py --model /path/to/sponza.obj`."""
    s = Scene()
    _std_materials(s)
    s.LoadModel(path, default_material)
    s.AddQuad((-2, 0, -7.5), (2, 0, -7.5), (2, 4, -7.5), (-2, 4, -7.5), "white-light")
    s.BuildBLAS(0, alpha)
    view = dict(origin=(-10.0, 10.0, 15.0), forward=(0.0, 0.0, 1.0), fov=110.0, aperture=0.1)
    return s, view


# ----------------------------------------------------------------------------------------- edge cases
def mixed_scene(alpha=1.0, textured=True):
    """Small scene exercising every primitive/material branch: triangles, spheres (diffuse, mirror,
    glass, emissive), a textured quad, two triangle lights."""
    s = Scene()
    _std_materials(s)
    if textured:
        yy, xx = np.mgrid[0:16, 0:16]
        tex = np.zeros((16, 16, 4), dtype=np.float32)
        tex[..., 0] = 0.2 + 0.6 * ((xx // 2 + yy // 2) % 2)
        tex[..., 1] = 0.3 + 0.04 * xx
        tex[..., 2] = 0.9 - 0.05 * yy
        s.AddTexture("checker", tex)
        s.AddTriangle((-4, 0, -4), (-4, 0, 4), (4, 0, 4), "checker", uv0=(0, 0), uv1=(0, 3), uv2=(3, 3))
        s.AddTriangle((4, 0, 4), (4, 0, -4), (-4, 0, -4), "checker", uv0=(3, 3), uv1=(3, 0), uv2=(0, 0))
    else:
        s.AddQuad((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4), "grey")
    s.AddQuad((-4, 0, -4), (4, 0, -4), (4, 5, -4), (-4, 5, -4), "white")
    s.AddQuad((-4, 0, -4), (-4, 5, -4), (-4, 5, 4), (-4, 0, 4), "red")
    s.AddTriangles(box_tris((1.2, 0, -1.8), (2.4, 1.6, -0.6)), "green")
    s.AddSphere((-1.5, 0.8, 0.2), 0.8, "mirror")
    s.AddSphere((0.4, 0.6, 1.0), 0.6, "white-glass")
    s.AddSphere((1.6, 0.45, 1.6), 0.45, "sand")
    s.AddSphere((-2.6, 2.8, -2.2), 0.35, "green-light")
    s.AddQuad((-1, 4.6, -1), (1, 4.6, -1), (1, 4.6, 1), (-1, 4.6, 1), "white-light")
    s.AddTriangle((3.2, 3.0, -3.0), (3.9, 3.0, -3.0), (3.9, 3.7, -2.2), "red-light")
    s.BuildBLAS(0, alpha)
    view = dict(origin=(2.2, 2.4, 5.2), forward=(0.32, 0.22, 0.92), fov=70.0, aperture=0.04)
    return s, view


def branch_scene(alpha=1.0):
    """A row of objects at eye level in a textured room, so that ONE thin horizontal band of the frame drives every shading
    branch of the reference within a few bounces: textured triangles (floor, back wall), a textured sphere (lat-long lookup),
    a thick absorbing glass box (inside rays, Beer's law, total internal reflection at its side faces), a glass sphere, a
    mirror sphere, a sphere light and an upright triangle light seen directly (emissive hit with lastSpecular set), both of
    them and a ceiling quad light sampled by NEE (emissive hit after a diffuse bounce returns black)."""
    s = Scene()
    _std_materials(s)
    s.AddMaterial("thick-glass", material(color=(1, 1, 1), dielectric=True, n1=1.0, n2=1.5, specular=0.04, absorption=(0.9, 0.25, 0.1)))
    yy, xx = np.mgrid[0:16, 0:16]
    tex = np.zeros((16, 16, 4), dtype=np.float32)
    tex[..., 0] = 0.25 + 0.6 * ((xx // 2 + yy // 2) % 2)
    tex[..., 1] = 0.3 + 0.04 * xx
    tex[..., 2] = 0.85 - 0.04 * yy
    s.AddTexture("checker", tex)
    yy, xx = np.mgrid[0:8, 0:32]
    tex2 = np.zeros((8, 32, 4), dtype=np.float32)
    tex2[..., 0] = 0.15 + 0.025 * xx
    tex2[..., 1] = 0.75 - 0.07 * yy
    tex2[..., 2] = 0.3 + 0.5 * (xx % 2)
    s.AddTexture("stripes", tex2)
    # textured floor and back wall (uv beyond 1: the fmod wrap), plain side walls and ceiling
    s.AddTriangle((-6, 0, -5), (-6, 0, 6), (6, 0, 6), "checker", uv0=(0, 0), uv1=(0, 4), uv2=(4, 4))
    s.AddTriangle((6, 0, 6), (6, 0, -5), (-6, 0, -5), "checker", uv0=(4, 4), uv1=(4, 0), uv2=(0, 0))
    s.AddTriangle((-6, 0, -5), (6, 0, -5), (6, 4, -5), "stripes", uv0=(0, 0), uv1=(2.5, 0), uv2=(2.5, 1.5))
    s.AddTriangle((6, 4, -5), (-6, 4, -5), (-6, 0, -5), "stripes", uv0=(2.5, 1.5), uv1=(0, 1.5), uv2=(0, 0))
    s.AddQuad((-6, 0, -5), (-6, 4, -5), (-6, 4, 6), (-6, 0, 6), "red")
    s.AddQuad((6, 0, -5), (6, 0, 6), (6, 4, 6), (6, 4, -5), "green")
    s.AddQuad((-6, 4, -5), (6, 4, -5), (6, 4, 6), (-6, 4, 6), "white")
    # the row of objects, centres at y = 1
    s.AddTriangles(box_tris((-4.6, 0.3, -0.7), (-3.2, 1.7, 0.7)), "thick-glass")
    s.AddSphere((-2.2, 1.0, 0.0), 0.7, "checker")
    s.AddSphere((-0.6, 1.0, 0.2), 0.7, "mirror")
    s.AddSphere((0.7, 1.0, -0.3), 0.35, "green-light")
    s.AddTriangle((1.4, 0.5, -1.0), (2.4, 0.5, -1.0), (1.9, 1.6, -1.0), "red-light")          # upright, faces +z (the camera)
    s.AddSphere((3.0, 1.0, 0.1), 0.65, "white-glass")
    s.AddSphere((4.5, 1.0, 0.0), 0.7, "sand")
    s.AddQuad((-1.2, 3.95, -1.2), (1.2, 3.95, -1.2), (1.2, 3.95, 1.2), (-1.2, 3.95, 1.2), "white-light")   # faces down
    s.BuildBLAS(0, alpha)
    view = dict(origin=(0.0, 1.0, 5.6), forward=(0.0, 0.0, 1.0), fov=80.0, aperture=0.03)
    return s, view


def two_blas_scene(alpha=0.0, n=24):
    """Two BLAS under a TLAS (shape of config 5): a displaced blob and a torus-like ring, each its own
    BuildBLAS call, SBVH alpha given; a glass sphere sits in the second BLAS."""
    s = Scene()
    _std_materials(s)

    def blob(U, V):
        th, ph = U * 2 * np.pi, V * np.pi
        r = 1.0 + 0.2 * np.sin(4 * th) * np.sin(ph) ** 2
        return -1.6 + r * np.sin(ph) * np.cos(th), 1.3 + r * np.cos(ph), r * np.sin(ph) * np.sin(th)
    s.AddTriangles(param_surface(blob, n, n), "sand")
    s.AddQuad((-8, 0, -8), (-8, 0, 8), (8, 0, 8), (8, 0, -8), "grey")
    s.AddQuad((-1.5, 5, -1.5), (1.5, 5, -1.5), (1.5, 5, 1.5), (-1.5, 5, 1.5), "white-light")
    s.BuildBLAS(0, alpha)
    start = s.num_prims

    def ring(U, V):
        th, ph = U * 2 * np.pi, V * 2 * np.pi
        R, r = 1.1, 0.35
        return 1.9 + (R + r * np.cos(ph)) * np.cos(th), 1.2 + r * np.sin(ph), (R + r * np.cos(ph)) * np.sin(th)
    s.AddTriangles(param_surface(ring, n, n // 2), "mirror")
    s.AddSphere((1.9, 1.2, 0.0), 0.55, "white-glass")
    s.BuildBLAS(start, alpha)
    view = dict(origin=(0.3, 2.6, 5.6), forward=(0.02, 0.22, 0.97), fov=65.0, aperture=0.03)
    return s, view


def config5_scene(alpha=0.0, decimate=1):
    """BASELINE config 5: robo-orb (35,600 tris) + terrarium_bot (40,012 tris), each its own BLAS under a TLAS, SBVH
    alpha (0 = full spatial splits), glass on the terrarium dome.  Geometry: magr_ray_tracer_amd/assets/*.npz (converted
    from the glTF files that ship with the reference; CC-BY-4.0, see assets/ATTRIBUTION.md)."""
    import os
    from . import gltf
    adir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
    s = Scene()
    _std_materials(s)

    def place(name, height, at, matmap, default):
        V, I, M, names = gltf.load_packed(os.path.join(adir, name))
        V = V.astype(np.float64)
        lo, hi = V.min(0), V.max(0)
        sc = height / (hi[1] - lo[1])
        V = (V - np.array([(lo[0] + hi[0]) / 2, lo[1], (lo[2] + hi[2]) / 2])) * sc + np.array(at)
        V = V.astype(np.float32)
        for mi, mn in enumerate(names):
            tri = I[M == mi][::decimate]
            if len(tri):
                s.AddTriangles(V[tri], matmap.get(mn, default))

    place("robo_orb.npz", 1.9, (-1.5, 0.0, 0.0), {"Coat": "mirror", "Coat_2": "mirror", "Butt": "red", "material": "green"}, "white")
    s.AddQuad((-9, 0, -9), (-9, 0, 9), (9, 0, 9), (9, 0, -9), "grey")
    s.AddQuad((-1.5, 5.5, -1.5), (1.5, 5.5, -1.5), (1.5, 5.5, 1.5), (-1.5, 5.5, 1.5), "white-light")
    s.BuildBLAS(0, alpha)
    start = s.num_prims
    place("terrarium_bot.npz", 2.4, (1.7, 0.0, 0.0), {"glass": "white-glass", "ground": "sand", "inside": "sand", "pipes": "red"}, "white")
    s.BuildBLAS(start, alpha)
    view = dict(origin=(0.3, 2.3, 5.4), forward=(0.03, 0.2, 0.98), fov=62.0, aperture=0.02)
    return s, view


def camera_for(view, width, height, focalLength=1.0):
    return make_camera(width, height, view["origin"], view["forward"], fov=view.get("fov", 110.0),
                       aperture=view.get("aperture", 0.1), focalLength=focalLength, type=view.get("type", 0))
