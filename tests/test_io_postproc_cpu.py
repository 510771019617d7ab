"""CPU suite for the 'next' rows: post-processing chain restatement, PNG writer, OBJ loader."""
import os
import struct
import zlib

import numpy as np

from magr_ray_tracer_amd import _lib as W, scenes
from magr_ray_tracer_amd.scene import Scene, material, save_png
from oracle.oracle_py import Oracle, postproc
from helpers import DEFAULT, bits_equal


def _accum(Wd=48, Hd=27, frames=3):
    s, view = scenes.cube_scene()
    sa = s.arrays()
    cam = scenes.camera_for(view, Wd, Hd)
    acc, *_ = Oracle(sa, Wd, Hd, **DEFAULT).render(cam, frames)
    return acc


def test_postproc_prep_only_is_divide_and_clamp():
    acc = _accum()
    f, b = postproc(acc, 3, vignette=0.0, gamma=1.0, chromatic=0.0)
    exp = np.minimum(acc[..., :3] * np.float32(1 / np.float32(3)), np.float32(1))
    assert bits_equal(f[..., :3], exp) and np.all(f[..., 3] == 1)
    assert np.array_equal(b[..., :3], (exp * np.float32(255)).astype(np.uint8)) and np.all(b[..., 3] == 255)


def test_postproc_stages_match_numpy_model():
    acc = _accum()
    base, _ = postproc(acc, 3, 0.0, 1.0, 0.0)
    # gamma (default 0.9, renderer.h:30)
    g, _ = postproc(acc, 3, 0.0, 0.9, 0.0)
    assert np.allclose(g[..., :3], np.minimum(np.power(base[..., :3], np.float32(0.9)), 1), rtol=2e-6, atol=1e-7)
    # vignetting darkens with distance from the centre and leaves the centre almost untouched
    v, _ = postproc(acc, 3, 0.8, 1.0, 0.0)
    H, Wd = acc.shape[:2]
    yy, xx = np.mgrid[0:H, 0:Wd]
    d = np.sqrt((xx / Wd - 0.5) ** 2 + (yy / H - 0.5) ** 2)
    t = np.clip(d, 0, 1)
    vig = 1 - t * t * (3 - 2 * t)
    exp = base[..., :3] + (base[..., :3] * vig[..., None] - base[..., :3]) * 0.8
    assert np.allclose(v[..., :3], exp, rtol=1e-5, atol=1e-6)
    # chromatic: red untouched, first column untouched, green/blue blended with the left neighbour
    c, _ = postproc(acc, 3, 0.0, 1.0, 0.1)
    assert bits_equal(c[..., 0], base[..., 0]) and bits_equal(c[:, 0], base[:, 0])
    exp_g = base[:, 1:, 1] * np.float32(0.9) + base[:, :-1, 1] * np.float32(0.1)
    assert np.allclose(c[:, 1:, 1], np.minimum(exp_g, 1), rtol=1e-6, atol=1e-7)


def test_png_writer_roundtrip(tmp_path):
    rng = np.random.default_rng(7)
    img = rng.random((13, 21, 4), dtype=np.float32) * 1.3     # some values above 1: clamped like SaveImageF
    p = tmp_path / "x.png"
    save_png(p, img)
    raw = p.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        crc = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0]
        assert crc == zlib.crc32(typ + data) & 0xffffffff
        chunks.setdefault(typ, b"")
        chunks[typ] += data
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (w, h, depth, ctype) == (21, 13, 8, 2)
    px = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(13, 1 + 21 * 3)
    assert np.all(px[:, 0] == 0)
    exp = (np.minimum(img[..., :3], 1) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(px[:, 1:].reshape(13, 21, 3), exp)


OBJ = """# unit quad + a triangle, with texcoords and a material library
mtllib m.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 2 0 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
usemtl textured
f 1/1 2/2 3/3 4/4
usemtl plain
f -1 2 3
"""
MTL = "newmtl textured\nmap_Kd checker.png\nnewmtl plain\nKd 1 0 0\n"


def test_obj_loader_conventions(tmp_path):
    (tmp_path / "m.obj").write_text(OBJ)
    (tmp_path / "m.mtl").write_text(MTL)
    s = Scene()
    s.AddMaterial("white", material(color=(.8, .8, .8)))
    tex = np.zeros((2, 2, 4), np.float32)
    s.AddTexture("checker.png", tex)                # LoadTexture names the material after the diffuse texture
    n = s.LoadModel(tmp_path / "m.obj", "white", pos=(10, 0, 0))
    assert n == 3                                   # quad -> 2 triangles (fan), + 1
    s.BuildBLAS(0, 1.0)
    sa = s.arrays()
    p = sa.prims
    assert np.all(p["objType"] == W.PRIM_TRIANGLE)
    # first fan triangle (v1,v2,v3) with the face's vertex order reversed, translated by pos
    assert np.allclose(p["v0"][0][:3], [11, 1, 0]) and np.allclose(p["v1"][0][:3], [11, 0, 0]) and np.allclose(p["v2"][0][:3], [10, 0, 0])
    # texcoords keep file order and get v -> 1-v
    assert np.allclose(p["uv0"][0], [0, 1]) and np.allclose(p["uv1"][0], [1, 1]) and np.allclose(p["uv2"][0], [1, 0])
    mats = sa.mats
    assert mats[p["matIdx"][0]]["texIdx"] == 0 and mats[p["matIdx"][2]]["texIdx"] == -1   # textured face / default material
    # negative index resolves to the last vertex
    assert np.allclose(p["v2"][2][:3], [12, 0, 1])
    # reversed winding flips the geometric normal (0,0,1) -> (0,0,-1)
    assert np.allclose(p["N"][0][:3], [0, 0, -1])


def test_gltf_reader_applies_node_transforms(tmp_path):
    import json
    from magr_ray_tracer_amd import gltf
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    idx = np.array([0, 1, 2], np.uint16)
    blob = pos.tobytes() + idx.tobytes() + b"\x00\x00"
    (tmp_path / "s.bin").write_bytes(blob)
    g = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
         "nodes": [{"children": [1], "translation": [10, 0, 0]}, {"mesh": 0, "scale": [2, 2, 2], "rotation": [0, 0, 0.70710678, 0.70710678]}],
         "meshes": [{"name": "tri", "primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]}],
         "materials": [{"name": "glass"}], "buffers": [{"uri": "s.bin", "byteLength": len(blob)}],
         "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}],
         "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                       {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}]}
    (tmp_path / "s.gltf").write_text(json.dumps(g))
    parts = gltf.load_gltf(str(tmp_path / "s.gltf"))
    assert len(parts) == 1 and parts[0]["material"] == "glass"
    # scale 2, rotate 90 deg about z, then translate x+10: (1,0,0) -> (10,2,0), (0,1,0) -> (8,0,0)
    assert np.allclose(parts[0]["vertices"], [[10, 0, 0], [10, 2, 0], [8, 0, 0]], atol=1e-5)
    gltf.pack(parts, str(tmp_path / "m.npz"), source="unit")
    V, I, M, names = gltf.load_packed(str(tmp_path / "m.npz"))
    assert V.shape == (3, 3) and I.tolist() == [[0, 1, 2]] and names == ["glass"]


def test_config5_scene_two_blas_from_converted_assets():
    s, view = scenes.config5_scene(alpha=1.0, decimate=8)      # plain SAH and 1/8 of the triangles: fast on CPU
    sa = s.arrays()
    assert len(sa.blas) == 2 and len(sa.tlas) == 4 and sa.tlas[0]["leftRight"] == (1 | (2 << 16))
    assert sa.blas["bvhIdx"][1] > 0
    glass = [i for i, m in enumerate(sa.mats) if m["isDielectric"]]
    assert len(glass) == 1 and (sa.prims["matIdx"] == glass[0]).sum() > 100      # the terrarium dome
    o = Oracle(sa, 64, 36, **DEFAULT)
    cam = scenes.camera_for(view, 64, 36)
    acc, _, e, c = o.render(cam, 1)
    assert e["tlas_visits"] > 0 and e["inst_visits"] > e["rays"] and acc[..., :3].mean() > 0.01
