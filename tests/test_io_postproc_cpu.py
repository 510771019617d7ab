"""CPU suite for the 'next' rows: post-processing chain restatement, PNG writer, OBJ loader."""
import os
import struct
import zlib

import numpy as np
import pytest

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
    assert n == 3                                   # quad -> 2 triangles (tinyobjloader: cut along the shorter diagonal, 1-3 on a tie), + 1
    s.BuildBLAS(0, 1.0)
    sa = s.arrays()
    p = sa.prims
    assert np.all(p["objType"] == W.PRIM_TRIANGLE)
    # first triangle (corners 1, 2, 4 of the square) with the face's vertex order reversed, translated by pos
    assert np.allclose(p["v0"][0][:3], [10, 1, 0]) and np.allclose(p["v1"][0][:3], [11, 0, 0]) and np.allclose(p["v2"][0][:3], [10, 0, 0])
    # texcoords keep file order and get v -> 1-v
    assert np.allclose(p["uv0"][0], [0, 1]) and np.allclose(p["uv1"][0], [1, 1]) and np.allclose(p["uv2"][0], [0, 0])
    mats = sa.mats
    assert mats[p["matIdx"][0]]["texIdx"] == 0 and mats[p["matIdx"][2]]["texIdx"] == -1   # textured face / default material
    # negative index resolves to the last vertex
    assert np.allclose(p["v2"][2][:3], [12, 0, 1])
    # reversed winding flips the geometric normal (0,0,1) -> (0,0,-1)
    assert np.allclose(p["N"][0][:3], [0, 0, -1])


def write_textured_obj(tmp_path):
    """A 2 x 1.5 quad with texcoords + a plain triangle, an MTL whose `textured` material names tex.png, and that 4x3 RGB PNG."""
    (tmp_path / "m.obj").write_text(OBJ.replace("m.mtl", "t.mtl"))
    (tmp_path / "t.mtl").write_text("newmtl textured\nKd 1 1 1\nmap_Kd tex.png\nnewmtl plain\nKd 1 0 0\n")
    pix = (np.arange(4 * 3 * 3).reshape(3, 4, 3) * 7 % 256).astype(np.uint8)
    (tmp_path / "tex.png").write_bytes(_png_bytes(pix, 2))
    return pix


def test_load_model_loads_the_mtl_textures(tmp_path):
    """Scene::LoadModel (scene.cpp:190-195): every MTL material with a diffuse texture has its image loaded into the atlas and a
    material named after the texture added BEFORE the faces, so a bare LoadModel("x.obj", "white") renders textured."""
    pix = write_textured_obj(tmp_path)
    s = Scene()
    s.AddMaterial("white", material(color=(.8, .8, .8)))
    n = s.LoadModel(tmp_path / "m.obj", "white")
    assert n == 3
    s.BuildBLAS(0, 1.0)
    sa = s.arrays()
    m = sa.mats[sa.prims["matIdx"][0]]
    assert (int(m["texIdx"]), int(m["texW"]), int(m["texH"])) == (0, 4, 3) and not m["isDielectric"]
    assert sa.prims["matIdx"][0] == sa.prims["matIdx"][1] != sa.prims["matIdx"][2]
    assert sa.mats[sa.prims["matIdx"][2]]["texIdx"] == -1                      # `plain` has no map_Kd: the default material
    assert len(sa.tex) == 12 and np.array_equal(sa.tex[:, :3], _stb_float(pix).reshape(-1, 3))
    # loading the same model again loads its images again, as the reference does (LoadTexture appends; the material of that name is replaced)
    s2 = Scene()
    s2.AddMaterial("white", material(color=(.8, .8, .8)))
    s2.LoadModel(tmp_path / "m.obj", "white")
    s2.LoadModel(tmp_path / "m.obj", "white", pos=(3, 0, 0))
    assert len(s2.texture_array()) == 24
    s2.BuildBLAS(0, 1.0)
    sa2 = s2.arrays()
    assert sa2.mats[sa2.prims["matIdx"][3]]["texIdx"] == 12 and sa2.mats[sa2.prims["matIdx"][0]]["texIdx"] == 0
    # forceDefaultMat: the images are still loaded (scene.cpp:190-195 does not look at the flag), no face uses them
    s3 = Scene()
    s3.AddMaterial("white", material(color=(.8, .8, .8)))
    s3.LoadModel(tmp_path / "m.obj", "white", forceDefaultMat=True)
    assert len(s3.texture_array()) == 12
    s3.BuildBLAS(0, 1.0)
    sa3 = s3.arrays()
    assert all(sa3.mats[m]["texIdx"] == -1 for m in sa3.prims["matIdx"])
    # a missing image is reported, not silently rendered untextured
    (tmp_path / "tex.png").unlink()
    s4 = Scene()
    s4.AddMaterial("white", material(color=(.8, .8, .8)))
    with pytest.raises(RuntimeError, match="tex.png"):
        s4.LoadModel(tmp_path / "m.obj", "white")


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


# ------------------------------------------------------------------ texture files (Scene::LoadTexture, image_io.cpp)
def _png_bytes(pix, ctype, depth=8, palette=None, filters=(0, 1, 2, 3, 4), interlace=False):
    """Encode `pix` (h, w, channels) as a PNG with the given colour type, cycling through the scanline filters; interlace=True
    writes the seven Adam7 passes (PNG 1.2 section 8.2)."""
    import struct
    import zlib
    pix = np.asarray(pix)
    h, w = pix.shape[:2]
    chan = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    pix = pix.reshape(h, w, chan)
    bpp = max(1, chan * depth // 8)

    def filtered(sub):
        sh, sw = sub.shape[:2]
        if depth == 16:
            rows = sub.astype(">u2").reshape(sh, sw * chan).view(np.uint8).reshape(sh, -1)
        elif depth == 8:
            rows = sub.astype(np.uint8).reshape(sh, sw * chan)
        else:   # packed samples, most significant first
            per = 8 // depth
            padded = np.zeros((sh, (sw + per - 1) // per * per), dtype=np.uint8)
            padded[:, :sw] = sub.reshape(sh, sw)
            rows = np.zeros((sh, padded.shape[1] // per), dtype=np.uint8)
            for k in range(per):
                rows |= (padded[:, k::per] << ((per - 1 - k) * depth)).astype(np.uint8)
        raw = bytearray()
        prev = np.zeros(rows.shape[1], dtype=np.int32)
        for y in range(sh):
            cur = rows[y].astype(np.int32)
            a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if bpp < cur.size else np.zeros_like(cur)
            c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if bpp < cur.size else np.zeros_like(cur)
            ft = filters[y % len(filters)]
            if ft == 0:
                f = cur
            elif ft == 1:
                f = cur - a
            elif ft == 2:
                f = cur - prev
            elif ft == 3:
                f = cur - ((a + prev) >> 1)
            else:
                p = a + prev - c
                pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
                f = cur - pred
            raw.append(ft)
            raw += bytes((f & 255).astype(np.uint8))
            prev = cur
        return raw

    if interlace:
        raw = bytearray()
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = pix[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filtered(sub)
    else:
        raw = filtered(pix)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, int(interlace)))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette.astype(np.uint8).reshape(-1)))
    z = zlib.compress(bytes(raw), 6)
    out += chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    return out


def _stb_float(v8):
    """stbi_loadf's ldr->hdr rule (lib/stb_image.h:1553,1849): (float)(pow(v / 255.0f, 2.2f) * 1.0f), pow in double."""
    x = (v8.astype(np.float32) / np.float32(255.0)).astype(np.float64)
    return np.power(x, np.float64(np.float32(2.2))).astype(np.float32)


def _texels(scene):
    return np.array(scene.texture_array())


def test_load_texture_png_variants(tmp_path):
    rng = np.random.default_rng(7)
    h, w = 13, 21
    cases = []
    rgb = rng.integers(0, 256, (h, w, 3))
    cases.append(("rgb8", _png_bytes(rgb, 2), rgb))
    rgba = rng.integers(0, 256, (h, w, 4))
    cases.append(("rgba8", _png_bytes(rgba, 6), rgba[..., :3]))
    g = rng.integers(0, 256, (h, w, 1))
    cases.append(("grey8", _png_bytes(g, 0), np.repeat(g, 3, axis=2)))
    ga = rng.integers(0, 256, (h, w, 2))
    cases.append(("greyalpha8", _png_bytes(ga, 4), np.repeat(ga[..., :1], 3, axis=2)))
    rgb16 = rng.integers(0, 65536, (h, w, 3))
    cases.append(("rgb16", _png_bytes(rgb16, 2, depth=16), rgb16 >> 8))
    pal = rng.integers(0, 256, (16, 3))
    idx = rng.integers(0, 16, (h, w, 1))
    cases.append(("pal4", _png_bytes(idx, 3, depth=4, palette=pal), pal[idx[..., 0]]))
    g2 = rng.integers(0, 4, (h, w, 1))
    cases.append(("grey2", _png_bytes(g2, 0, depth=2), np.repeat(g2 * 85, 3, axis=2)))
    s = Scene()
    off = 0
    for k, (name, data, expect8) in enumerate(cases):
        f = tmp_path / (name + ".png")
        f.write_bytes(data)
        mi = s.LoadTexture(f, name)
        t = _texels(s)
        m = s.material_array()[mi]
        assert (int(m["texIdx"]), int(m["texW"]), int(m["texH"])) == (off, w, h), name
        got = t[off:off + w * h].reshape(h, w, 4)
        assert np.array_equal(got[..., :3].view(np.uint32), _stb_float(expect8).view(np.uint32)), name
        assert (got[..., 3] == 0).all()
        off += w * h


def test_load_texture_roundtrip_of_save_png(tmp_path):
    """A file written by SavePNG (stored deflate blocks) is read back to the bytes SaveImageF's rule produced."""
    rng = np.random.default_rng(3)
    img = rng.random((9, 17, 4), dtype=np.float32) * 1.2
    f = tmp_path / "frame.png"
    save_png(f, img)
    s = Scene()
    s.LoadTexture(f, "frame")
    b = (np.minimum(img[..., :3], 1.0) * 255).astype(np.uint8)
    got = _texels(s).reshape(9, 17, 4)[..., :3]
    assert np.array_equal(got.view(np.uint32), _stb_float(b).view(np.uint32))


def test_load_texture_hdr_and_tga(tmp_path):
    rng = np.random.default_rng(11)
    h, w = 5, 12
    rgbe = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
    rgbe[0, 0, 3] = 0                                        # e == 0 -> black
    rgbe[2, 3:9] = rgbe[2, 3]                                # a run, so that the RLE path really codes runs
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)
    flat = head + rgbe.tobytes()
    rle = bytearray(head)
    for y in range(h):
        rle += bytes([2, 2, w >> 8, w & 255])
        for k in range(4):
            comp = rgbe[y, :, k]
            x = 0
            while x < w:
                n = 1
                while x + n < w and n < 127 and comp[x + n] == comp[x]:
                    n += 1
                if n >= 3:
                    rle += bytes([128 + n, int(comp[x])])
                else:
                    n = 1
                    while x + n < w and n < 128 and not (x + n + 2 < w and comp[x + n] == comp[x + n + 1] == comp[x + n + 2]):
                        n += 1
                    rle += bytes([n]) + comp[x:x + n].tobytes()
                x += n
    expect = np.where(rgbe[..., 3:] == 0, np.float32(0),
                      rgbe[..., :3].astype(np.float32) * np.ldexp(np.float32(1.0), rgbe[..., 3:].astype(np.int32) - 136).astype(np.float32))
    for name, data in (("flat.hdr", flat), ("rle.hdr", bytes(rle))):
        f = tmp_path / name
        f.write_bytes(data)
        s = Scene()
        s.LoadTexture(f, "sky")
        got = _texels(s).reshape(h, w, 4)
        assert np.array_equal(got[..., :3].view(np.uint32), expect.astype(np.float32).view(np.uint32)), name
    # TGA: 24-bit bottom-up uncompressed and 32-bit top-down run-length coded
    bgr = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    tga = bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, w & 255, w >> 8, h & 255, h >> 8, 24, 0]) + bgr.tobytes()
    f = tmp_path / "a.tga"
    f.write_bytes(tga)
    s = Scene()
    s.LoadTexture(f, "a")
    got = _texels(s).reshape(h, w, 4)[..., :3]
    assert np.array_equal(got.view(np.uint32), _stb_float(bgr[::-1, :, ::-1]).view(np.uint32))
    bgra = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
    bgra[1, 2:7] = bgra[1, 2]
    body = bytearray()
    flatpix = bgra.reshape(-1, 4)
    i = 0
    while i < flatpix.shape[0]:
        n = 1
        while i + n < flatpix.shape[0] and n < 128 and (flatpix[i + n] == flatpix[i]).all():
            n += 1
        if n > 1:
            body += bytes([0x80 | (n - 1)]) + flatpix[i].tobytes()
        else:
            body += bytes([0]) + flatpix[i].tobytes()
        i += n
    tga = bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, w & 255, w >> 8, h & 255, h >> 8, 32, 0x28]) + bytes(body)
    f = tmp_path / "b.tga"
    f.write_bytes(tga)
    s = Scene()
    s.LoadTexture(f, "b")
    got = _texels(s).reshape(h, w, 4)[..., :3]
    assert np.array_equal(got.view(np.uint32), _stb_float(bgra[:, :, 2::-1]).view(np.uint32))


def _loaded_bytes(path):
    """LoadTexture(path) mapped back to 8-bit levels through the inverse of stbi_loadf's rule (exact: the rule is a 256-entry table)."""
    s = Scene()
    mi = s.LoadTexture(path, "t")
    m = s.material_array()[mi]
    w, h = int(m["texW"]), int(m["texH"])
    tex = _texels(s)[:, :3]
    lut = _stb_float(np.arange(256))
    idx = np.clip(np.searchsorted(lut, tex.ravel()), 0, 255)
    assert np.array_equal(lut[idx], tex.ravel())
    return idx.reshape(h, w, 3).astype(np.int32)


def test_load_texture_jpeg_against_libjpeg(tmp_path):
    """Sanity check against an independent decoder (Pillow's libjpeg-turbo): baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0 / grey,
    optimised Huffman tables, restart markers, odd sizes - within the few levels by which conforming decoders differ (IDCT, chroma
    interpolation and colour rounding are not fixed by T.81).  The reconstruction follows the reference's stb_image bit for bit
    (tests/test_ref_io_cpu.py pins that); stb_image weights the LAST chroma sample pair of an odd-width row the other way round
    (lib/stb_image.h:3442), so the last column is left out here."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:157, 0:203]
    base = np.stack([(np.sin(xx / 9.0) + 1) * 100 + rng.integers(0, 20, xx.shape), (np.cos(yy / 7.0) + 1) * 90 + 30,
                     ((xx + yy) % 64) * 3 + rng.integers(0, 30, xx.shape)], -1).clip(0, 255).astype(np.uint8)
    colour, grey = Image.fromarray(base), Image.fromarray(base[..., 0])
    cases = [("base420", colour, dict(quality=85, subsampling=2)), ("base444", colour, dict(quality=90, subsampling=0)),
             ("base422", colour, dict(quality=75, subsampling=1)), ("prog420", colour, dict(quality=85, subsampling=2, progressive=True)),
             ("prog444", colour, dict(quality=92, subsampling=0, progressive=True)), ("opt", colour, dict(quality=60, subsampling=2, optimize=True)),
             ("rst", colour, dict(quality=85, subsampling=2, restart_marker_blocks=3)),
             ("grey", grey, dict(quality=80)), ("greyprog", grey, dict(quality=80, progressive=True)),
             ("tiny", colour.crop((0, 0, 5, 3)), dict(quality=90, subsampling=2))]
    for name, im, kw in cases:
        f = tmp_path / (name + ".jpg")
        im.save(f, "JPEG", **kw)
        ref = np.asarray(Image.open(f).convert("RGB")).astype(np.int32)
        got = _loaded_bytes(f)
        assert got.shape == ref.shape, name
        d = np.abs(got - ref)[:, :-1] if got.shape[1] > 1 else np.abs(got - ref)
        assert d.max() <= 5 and d.mean() < 0.3, (name, int(d.max()), float(d.mean()))


def test_load_texture_png_against_pillow(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(9)
    rgb = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    for name, im in (("rgb", Image.fromarray(rgb)), ("rgba", Image.fromarray(np.dstack([rgb, rgb[..., :1]]))),
                     ("pal", Image.fromarray(rgb).quantize(17)), ("grey", Image.fromarray(rgb[..., 0])),
                     ("bilevel", Image.fromarray(rgb[..., 0] > 127))):
        f = tmp_path / (name + ".png")
        im.save(f, "PNG", optimize=(name != "rgb"))
        ref = np.asarray(Image.open(f).convert("RGB")).astype(np.int32)
        assert np.array_equal(_loaded_bytes(f), ref), name


def test_load_texture_errors(tmp_path):
    s = Scene()
    with pytest.raises(Exception, match="no BLAS"):     # an empty scene is an error, not a crash
        s.arrays()
    with pytest.raises(Exception):
        s.LoadTexture(tmp_path / "missing.png", "x")
    f = tmp_path / "photo.jpg"
    f.write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 64)
    with pytest.raises(Exception, match="JPEG"):
        s.LoadTexture(f, "x")
    # regression (found by tools/fuzz_image_io.cpp under AddressSanitizer): a DHT segment whose code lengths over-subscribe the code space
    dht = bytes([0xff, 0xc4, 0x00, 0x16, 0x00, 3] + [0] * 15 + [1, 2, 3])
    f = tmp_path / "badhuff.jpg"
    f.write_bytes(b"\xff\xd8" + dht + b"\xff\xd9")
    with pytest.raises(Exception, match="Huffman"):
        s.LoadTexture(f, "x")
    # a frame header that claims 65535 x 65535 pixels is a corrupt file, not an allocation request
    f = tmp_path / "huge.jpg"
    f.write_bytes(b"\xff\xd8" + bytes([0xff, 0xc0, 0x00, 0x0b, 8, 0xff, 0xff, 0xff, 0xff, 1, 1, 0x11, 0]) + b"\xff\xd9")
    with pytest.raises(Exception, match="64 Mpixel"):
        s.LoadTexture(f, "x")
    f = tmp_path / "cut.png"
    f.write_bytes(_png_bytes(np.zeros((4, 4, 3), dtype=np.int64), 2)[:60])
    with pytest.raises(Exception):
        s.LoadTexture(f, "x")


def test_model_scene_hook_renders_an_obj_file(tmp_path):
    """scenes.model_scene (bench.py --model): an OBJ + MTL + texture through LoadModel, the reference's light quad (scene.cpp:67), one
    BLAS - and the oracle renders it (the hook for a real sponza.obj, which the reference checkout does not contain)."""
    pix = write_textured_obj(tmp_path)
    s, view = scenes.model_scene(str(tmp_path / "m.obj"))
    sa = s.arrays(bvh4=False)
    assert len(sa.prims) == 3 + 2 and len(sa.lights) == 2 and len(sa.tex) == pix.shape[0] * pix.shape[1]
    view.update(origin=(0.5, 0.5, 3.0), forward=(0.0, 0.0, 1.0), fov=60.0)       # the camera looks along -forward: at the model's quad in the z = 0 plane
    cam = scenes.camera_for(view, 32, 18)
    o = Oracle(sa, 32, 18)
    acc, _, e, _ = o.render(cam, 2)
    assert e["rays"] > 32 * 18 and np.isfinite(acc).all() and acc[..., :3].max() > 0
