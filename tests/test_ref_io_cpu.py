"""Asset readers against the REFERENCE's own readers.

The reference reads textures through its vendored lib/stb_image.h (LoadImageF, template/template.cpp:1613-1627) and models through its
vendored src/tiny_obj_loader.h (Scene::LoadModel, src/scene.cpp:178-243).  oracle/build_ref.sh compiles those headers where they lie,
behind oracle/ref_io_runner.cpp, into oracle/_ref/libref_io.so (tests/ref_io.py binds it).  Two kinds of test:

  * live (need libref_io.so; it is built in the container that holds /root/reference and travels to the GPU box): the host library's
    LoadTexture / LoadModel / SavePNG against the compiled reference on the reference's own asset files (where present), on synthetic
    image files of every variant and on randomized OBJ + MTL files - bit for bit;
  * golden (run anywhere): tests/golden/assets_ref.npz holds small input FILES (bytes) with what the compiled reference made of them
    (tests/golden/make_io_golden.py wrote it); the host library has to reproduce those outputs.

One documented difference is not compared: for 1- and 2-channel images the reference's LoadImageF indexes past the pixel
(template.cpp:1621-1623: undefined at the end of the buffer); the host library expands grey to r = g = b, which is checked against
stb_image's own 8-bit grey decode instead."""
import glob
import os

import numpy as np
import pytest

import ref_io
from magr_ray_tracer_amd import _lib as W
from magr_ray_tracer_amd.scene import Scene, _view, material, save_png
from test_io_postproc_cpu import _png_bytes, _stb_float

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "assets_ref.npz")
ASSETS = "/root/reference/assets"
live = pytest.mark.skipif(not ref_io.available(), reason="oracle/_ref/libref_io.so not built (needs /root/reference)")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def load_texture(path):
    """Scene::LoadTexture -> (h, w, 3) float32 texels."""
    s = Scene()
    mi = s.LoadTexture(str(path), "t")
    m = s.material_array()[mi]
    return np.array(s.texture_array())[:, :3].reshape(int(m["texH"]), int(m["texW"]), 3)


def check_image(path, label=None):
    """LoadTexture(path) == the reference's LoadImageF(path) bit for bit (>= 3 channels), or == stb's grey levels (1-2 channels)."""
    label = label or os.path.basename(str(path))
    ours = load_texture(path)
    ref, c = ref_io.load_image_f(path)
    assert ours.shape == ref.shape, (label, ours.shape, ref.shape)
    if c >= 3:
        assert np.array_equal(bits(ours), bits(ref)), (label, c, float(np.abs(ours - ref).max()))
    else:
        g = ref_io.load_image_u8(path)[..., 0]
        want = _stb_float(g)
        for k in range(3):
            assert np.array_equal(bits(ours[..., k]), bits(want)), (label, c, k)
    return c


# ------------------------------------------------------------------------------------------------ textures
@live
def test_reference_texture_assets_decode_like_stb_image():
    """Every PNG / JPEG the reference ships (sponza's 21 JPEG textures among them) through Scene::LoadTexture and through the
    reference's LoadImageF."""
    files = sorted(set(glob.glob(ASSETS + "/**/*.png", recursive=True) + glob.glob(ASSETS + "/**/*.jpg", recursive=True) +
                       glob.glob(ASSETS + "/**/*.JPG", recursive=True)))
    if not files:
        pytest.skip("reference assets not present on this machine")
    # the 2048 x 2048 glTF textures are all the same PNG flavour: two of them are enough
    big = [f for f in files if "terrarium_bot" in f or "robo-orb" in f]
    files = [f for f in files if f not in big] + big[:2]
    channels = {}
    for f in files:
        channels[os.path.relpath(f, ASSETS)] = check_image(f)
    assert sum(1 for f in channels if f.lower().endswith(".jpg")) >= 20 and 1 in channels.values() and 4 in channels.values()


@live
def test_png_variants_decode_like_stb_image(tmp_path):
    rng = np.random.default_rng(3)
    n = 0
    for w, h in ((1, 1), (7, 5), (33, 18)):
        for ctype, chan in ((0, 1), (2, 3), (4, 2), (6, 4)):
            for depth in (8, 16):
                pix = rng.integers(0, 1 << depth, (h, w, chan))
                f = tmp_path / f"c{ctype}_d{depth}_{w}x{h}.png"
                f.write_bytes(_png_bytes(pix, ctype, depth))
                check_image(f); n += 1
        for depth in (1, 2, 4):
            pix = rng.integers(0, 1 << depth, (h, w, 1))
            f = tmp_path / f"g{depth}_{w}x{h}.png"
            f.write_bytes(_png_bytes(pix, 0, depth))
            check_image(f); n += 1
            pal = rng.integers(0, 256, (1 << depth, 3))
            f = tmp_path / f"p{depth}_{w}x{h}.png"
            f.write_bytes(_png_bytes(pix, 3, depth, palette=pal))
            check_image(f); n += 1
        pix = rng.integers(0, 256, (h, w, 1))
        f = tmp_path / f"p8_{w}x{h}.png"
        f.write_bytes(_png_bytes(pix, 3, 8, palette=rng.integers(0, 256, (256, 3))))
        check_image(f); n += 1
    assert n == 3 * (8 + 6 + 1)
    # interlaced (Adam7) files, sizes below and above one 8 x 8 pattern
    for w, h in ((1, 1), (2, 3), (5, 5), (8, 8), (9, 17), (33, 18), (64, 3)):
        for ctype, chan, depth in ((2, 3, 8), (6, 4, 8), (2, 3, 16), (0, 1, 8), (0, 1, 1), (0, 1, 4), (4, 2, 8)):
            f = tmp_path / f"i{ctype}_{depth}_{w}x{h}.png"
            f.write_bytes(_png_bytes(rng.integers(0, 1 << depth, (h, w, chan)), ctype, depth, interlace=True))
            check_image(f)
        f = tmp_path / f"ip2_{w}x{h}.png"
        f.write_bytes(_png_bytes(rng.integers(0, 4, (h, w, 1)), 3, 2, palette=rng.integers(0, 256, (4, 3)), interlace=True))
        check_image(f)


def jpeg_cases():
    """(name, PIL image, save options): baseline / progressive, every chroma layout Pillow writes, optimised tables, restart
    intervals, grey, sizes that are not whole MCUs, one-sample-wide chroma rows."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:61, 0:83]
    base = np.stack([(np.sin(xx / 9.0) + 1) * 100 + rng.integers(0, 20, xx.shape), (np.cos(yy / 7.0) + 1) * 90 + 30,
                     ((xx + yy) % 64) * 3 + rng.integers(0, 30, xx.shape)], -1).clip(0, 255).astype(np.uint8)
    noise = rng.integers(0, 256, (37, 29, 3)).astype(np.uint8)          # saturates the IDCT: clamping and 16-bit wrap paths
    colour, grey, hard = Image.fromarray(base), Image.fromarray(base[..., 0]), Image.fromarray(noise)
    cases = [("base420", colour, dict(quality=85, subsampling=2)), ("base444", colour, dict(quality=90, subsampling=0)),
             ("base422", colour, dict(quality=75, subsampling=1)), ("prog420", colour, dict(quality=85, subsampling=2, progressive=True)),
             ("prog444", colour, dict(quality=92, subsampling=0, progressive=True)), ("prog422", colour, dict(quality=70, subsampling=1, progressive=True)),
             ("opt", colour, dict(quality=60, subsampling=2, optimize=True)), ("rst", colour, dict(quality=85, subsampling=2, restart_marker_blocks=3)),
             ("q100", hard, dict(quality=100, subsampling=2)), ("q5", hard, dict(quality=5, subsampling=0)), ("q30_422", hard, dict(quality=30, subsampling=1)),
             ("grey", grey, dict(quality=80)), ("greyprog", grey, dict(quality=80, progressive=True))]
    for w, h in ((1, 1), (2, 2), (3, 1), (1, 5), (5, 3), (16, 16), (17, 9), (8, 33)):
        for sub in (0, 1, 2):
            cases.append((f"s{sub}_{w}x{h}", colour.crop((10, 10, 10 + w, 10 + h)), dict(quality=88, subsampling=sub)))
    return cases


@live
def test_jpeg_variants_decode_like_stb_image(tmp_path):
    for name, im, kw in jpeg_cases():
        f = tmp_path / (name + ".jpg")
        im.save(f, "JPEG", **kw)
        check_image(f, name)


@live
def test_tga_and_hdr_decode_like_stb_image(tmp_path):
    rng = np.random.default_rng(11)
    w, h = 13, 7
    import struct
    for name, bpp, origin_top in (("t24", 24, False), ("t32", 32, True), ("t24top", 24, True)):
        pix = rng.integers(0, 256, (h, w, bpp // 8)).astype(np.uint8)
        hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, bpp, (0x20 if origin_top else 0) | (8 if bpp == 32 else 0))
        f = tmp_path / (name + ".tga")
        f.write_bytes(hdr + pix.tobytes())
        check_image(f)
    # grey (type 3) and run-length coded true colour (type 10) / grey (type 11)
    g = rng.integers(0, 256, (h, w)).astype(np.uint8)
    f = tmp_path / "grey.tga"
    f.write_bytes(struct.pack("<BBBHHBHHHHBB", 0, 0, 3, 0, 0, 0, 0, 0, w, h, 8, 0) + g.tobytes())
    check_image(f)
    for name, typ, bpp in (("rle24", 10, 24), ("rle32", 10, 32), ("rle8", 11, 8)):
        pix = rng.integers(0, 256, (h * w, bpp // 8)).astype(np.uint8)
        pix[20:50] = pix[20]
        body, i = bytearray(), 0
        while i < len(pix):                                    # packets may cross scanlines, as the format allows
            run = 1
            while i + run < len(pix) and run < 128 and np.array_equal(pix[i + run], pix[i]):
                run += 1
            if run >= 2:
                body += bytes([128 + run - 1]) + pix[i].tobytes(); i += run
            else:
                n = min(int(rng.integers(1, 6)), len(pix) - i)
                body += bytes([n - 1]) + pix[i:i + n].tobytes(); i += n
        f = tmp_path / (name + ".tga")
        f.write_bytes(struct.pack("<BBBHHBHHHHBB", 0, 0, typ, 0, 0, 0, 0, 0, w, h, bpp, 0x20 | (8 if bpp == 32 else 0)) + bytes(body))
        check_image(f)
    # Radiance RGBE, flat (non-RLE) scanlines
    rgbe = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
    rgbe[0, 0, 3] = 0
    f = tmp_path / "flat.hdr"
    f.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode() + rgbe.tobytes())
    check_image(f)
    # new-style run-length coded scanlines (width 8..32767): per channel, runs (128 + n, value) and literal packets (n, bytes)
    w2 = 40
    img = rng.integers(0, 256, (h, w2, 4)).astype(np.uint8)
    img[:, 5:25, 1] = 77                                     # something worth a run
    body = bytearray()
    for y in range(h):
        body += bytes([2, 2, w2 >> 8, w2 & 255])
        for ch in range(4):
            row, x = img[y, :, ch], 0
            while x < w2:
                run = 1
                while x + run < w2 and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    body += bytes([128 + run, int(row[x])]); x += run
                else:
                    n = min(int(rng.integers(1, 9)), w2 - x)
                    body += bytes([n]) + row[x:x + n].tobytes(); x += n
    f = tmp_path / "rle.hdr"
    f.write_bytes(b"#?RADIANCE\n# made by a test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + f"-Y {h} +X {w2}\n".encode() + bytes(body))
    check_image(f)


@live
def test_save_png_reads_back_like_stbi_write_png(tmp_path):
    """SavePNG (SaveImageF's byte rule) and the reference's stbi_write_png hold the same pixels (the deflate streams differ)."""
    rng = np.random.default_rng(2)
    img = rng.uniform(0.0, 1.3, (19, 23, 4)).astype(np.float32)
    save_png(tmp_path / "ours.png", img)
    c = np.minimum(img[..., :3], np.float32(1.0))
    rgb8 = (c * np.float32(255)).astype(np.int32).astype(np.uint8)      # template.cpp:1635-1641: clamp to 1, (uchar)(c * 255)
    ref_io.write_png(tmp_path / "ref.png", rgb8)
    a, b = ref_io.load_image_u8(tmp_path / "ours.png"), ref_io.load_image_u8(tmp_path / "ref.png")
    assert a.shape == b.shape == (19, 23, 3) and np.array_equal(a, b)


# ------------------------------------------------------------------------------------------------ models
NUMBERS = ["0", "1", "-2", "+3", ".5", "-.25", "+.75", "1.", "-0.0", "1e3", "2E-3", "-4.5e+2", "1e38", "1e39", "1e-38", "1e-46", "7e-324",
           "0.123456789012345678", "123456789.987654321", "3.14159265358979", "1e", "abc", "nan", "inf", "--1", "1.5.2", "0x10", "1e400", "1e-400",
           "16777217", "0.1", "0.2", "0.3", "1.0000001", "0.99999994", "33554433.5"]


def number(rng):
    k = rng.integers(0, 10)
    x = rng.uniform(-3, 3)
    if k == 0:
        return NUMBERS[rng.integers(0, len(NUMBERS))]
    if k == 1:
        return f"{x:.{rng.integers(0, 18)}f}"
    if k == 2:
        return f"{x * 10.0 ** rng.integers(-12, 12):.{rng.integers(0, 10)}e}"
    if k == 3:
        return repr(float(np.float32(x)))
    return f"{x:.6f}"


def random_obj(rng, with_mtl=True):
    """An OBJ text (and its MTL text) exercising tinyobjloader's reader: number spellings, index forms, polygons of 3-8 corners
    (planar, convex and not, and arbitrary), groups / objects / materials, blank and comment lines, tabs, all three line endings."""
    nv, nt = int(rng.integers(8, 40)), int(rng.integers(0, 12))
    sep = lambda: " " * int(rng.integers(1, 3)) if rng.integers(0, 5) else "\t"
    L = ["# random", ""]
    mats = []
    if with_mtl:
        L.append("mtllib" + " " + "m.mtl")
        mats = ["plain", "texA", "texB", "twoWords", "dupe", "notThere"]
    verts = []
    for i in range(nv):
        if i >= 8 and rng.integers(0, 3) == 0:      # a planar polygon's worth of vertices now and then
            k = int(rng.integers(5, 9)); r = rng.uniform(0.3, 1.0, k) if rng.integers(0, 2) else np.ones(k)
            cx, cy, cz = rng.uniform(-2, 2, 3)
            for j in range(k):
                a = 2 * np.pi * j / k
                verts.append((f"{cx + r[j] * np.cos(a):.6f}", f"{cy + r[j] * np.sin(a):.6f}", f"{cz:.6f}"))
        else:
            verts.append((number(rng), number(rng), number(rng)))
    for v in verts:
        L.append("v" + sep() + sep().join(v) + (sep() + "0.5 0.25 0.125" if rng.integers(0, 9) == 0 else "") + (" " if rng.integers(0, 6) == 0 else ""))
    for _ in range(nt):
        L.append("vt" + sep() + number(rng) + sep() + number(rng) + (sep() + "0" if rng.integers(0, 4) == 0 else ""))
    L.append("vn 0 0 1")
    nv = len(verts)

    def corner(i):
        form = rng.integers(0, 5) if nt else rng.integers(0, 2) * 3
        vi = str(i + 1) if rng.integers(0, 4) else str(i - nv)
        ti = int(rng.integers(0, nt)) if nt else 0
        ts = str(ti + 1) if rng.integers(0, 4) else str(ti - nt)
        return vi if form == 0 else (f"{vi}/{ts}" if form in (1, 2) else (f"{vi}//1" if form == 3 else f"{vi}/{ts}/1"))
    for _ in range(int(rng.integers(5, 30))):
        r = rng.integers(0, 12)
        if r == 0:
            L.append("g" + sep() + f"grp{rng.integers(0, 5)}")
        elif r == 1:
            L.append("o" + sep() + f"obj{rng.integers(0, 5)}")
        elif r == 2 and mats:
            L.append("usemtl" + sep() + mats[rng.integers(0, len(mats))])
        elif r == 3:
            L.append(["s 1", "s off", "# c", "", "l 1 2", "   "][rng.integers(0, 6)])
        else:
            k = int(rng.choice([3, 3, 3, 4, 4, 4, 5, 6, 7, 8, 2]))
            if k >= 5 and rng.integers(0, 2):
                start = int(rng.integers(0, max(1, nv - k)))
                idx = list(range(start, start + k))                # consecutive: often one of the planar rings
            else:
                idx = [int(x) for x in rng.integers(0, nv, k)]
            L.append(("  " if rng.integers(0, 8) == 0 else "") + "f" + sep() + sep().join(corner(i) for i in idx) + ("  " if rng.integers(0, 5) == 0 else ""))
    eol = ["\n", "\r\n", "\r"][rng.integers(0, 3)]
    obj = eol.join(L) + (eol if rng.integers(0, 2) else "")
    mtl = ("# materials\nnewmtl plain\nKd 1 0 0\n\nnewmtl texA\nKd 1 1 1\nmap_Kd a.png\nnewmtl texB\n\tmap_Kd -s 1 1 1 -clamp on b.png  \n"
           "newmtl twoWords\nmap_Kd -bm 0.5 two words.png\nnewmtl dupe\nmap_Kd a.png\nmap_Bump bump.png\n")
    return obj, mtl


TEX = {"a.png": 2, "b.png": 3, "two words.png": 4}      # texture -> width (height 1): the texel count identifies the image


def write_model(d, obj, mtl):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "m.obj"), "w", newline="") as f:
        f.write(obj)
    with open(os.path.join(d, "m.mtl"), "w", newline="") as f:
        f.write(mtl)
    for name, w in TEX.items():
        pix = (np.arange(w * 3).reshape(1, w, 3) * 17 + w).astype(np.uint8)
        with open(os.path.join(d, name), "wb") as f:
            f.write(_png_bytes(pix, 2))
    return os.path.join(d, "m.obj")


def load_model(path, pos=(0.0, 0.0, 0.0), force=False):
    s = Scene()
    s.AddMaterial("white", material(color=(1, 1, 1)))
    n = s.LoadModel(path, "white", pos=pos, forceDefaultMat=force)
    prims = np.array(_view(s._lib.rth_primitives, s._h, W.Primitive))
    assert n == len(prims)
    return dict(verts=np.stack([prims["v0"][:, :3], prims["v1"][:, :3], prims["v2"][:, :3]], axis=1) if n else np.zeros((0, 3, 3), np.float32),
                uvs=np.stack([prims["uv0"], prims["uv1"], prims["uv2"]], axis=1) if n else np.zeros((0, 3, 2), np.float32),
                mat=prims["matIdx"].astype(np.int64), mats=np.array(s.material_array()), atlas=np.array(s.texture_array()))


def check_model(ours, ref, label):
    """`ours`: load_model(); `ref`: what the reference's LoadModel hands to AddTriangle (ref_io.obj_load or a golden)."""
    assert len(ours["verts"]) == len(ref["verts"]), (label, len(ours["verts"]), len(ref["verts"]))
    assert np.array_equal(bits(ours["verts"]), bits(ref["verts"])), (label, "vertices")
    assert np.array_equal(bits(ours["uvs"]), bits(ref["uvs"])), (label, "texcoords")
    names = list(ref["tex_names"])
    for i, t in enumerate(ref["tex"]):                      # the material handed to AddTriangle
        m = ours["mats"][ours["mat"][i]]
        want = names[int(t)]
        if want == "white":
            assert int(m["texIdx"]) == -1, (label, i)
        else:
            assert (int(m["texW"]), int(m["texH"])) == (TEX[want], 1), (label, i, want)
    # scene.cpp:190-195: one LoadTexture per MTL material with a diffuse texture, in MTL order (duplicates included)
    want_texels = sum(TEX[d] for d in ref["diffuse"] if d)
    assert len(ours["atlas"]) == want_texels, (label, len(ours["atlas"]), want_texels)


@live
def test_load_model_matches_tinyobjloader_on_random_files(tmp_path):
    rng = np.random.default_rng(20261004)
    polygons = 0
    for k in range(150):
        obj, mtl = random_obj(rng, with_mtl=k % 5 != 0)
        path = write_model(str(tmp_path / f"m{k}"), obj, mtl)
        pos = (0.0, 0.0, 0.0) if k % 3 else tuple(float(x) for x in rng.uniform(-5, 5, 3))
        force = k % 7 == 0
        try:
            r = ref_io.obj_load(path, "white", pos, force)
        except RuntimeError as e:                            # the reader refuses the file (e.g. a zero index): so must the host library
            assert "TinyObjReader" in str(e)
            with pytest.raises(RuntimeError):
                load_model(path, pos, force)
            continue
        r["diffuse"] = [d for _, d in r["materials"]]
        check_model(load_model(path, pos, force), r, f"file {k}")
        polygons += len(r["verts"])
    assert polygons > 2000


@live
def test_load_model_quad_and_polygon_rules(tmp_path):
    """The cases worked out by hand: a quad is cut along its shorter diagonal (0-2 only when strictly shorter), a concave pentagon is
    ear-clipped, faces of one or two corners are dropped, a zero index fails the file."""
    cases = {"square": ["v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "f 1 2 3 4"],
             "kite02": ["v 0 0 0", "v 2 -1 0", "v 1 0 0", "v 2 1 0", "f 1 2 3 4"],
             "kite13": ["v 0 0 0", "v 1 -3 0", "v 5 0 0", "v 1 3 0", "f 1 2 3 4"],
             "concave": ["v 0 0 0", "v 2 0 0", "v 2 2 0", "v 1 0.5 0", "v 0 2 0", "f 1 2 3 4 5"],
             "degenerate": ["v 0 0 0", "v 1 0 0", "v 0 1 0", "f 1 2", "f 1", "f 1 2 3"],
             "collinear_hexagon": ["v 0 0 0", "v 1 0 0", "v 2 0 0", "v 2 1 0", "v 1 1 0", "v 0 1 0", "f 1 2 3 4 5 6"]}
    for name, lines in cases.items():
        path = write_model(str(tmp_path / name), "\n".join(lines) + "\n", "newmtl x\n")
        r = ref_io.obj_load(path, "white")
        r["diffuse"] = []
        check_model(load_model(path), r, name)
    sq = load_model(str(tmp_path / "square" / "m.obj"))["verts"]
    # equal diagonals: triangles (0,1,3) and (1,2,3), each stored with its vertex list reversed
    assert sq.tolist() == [[[0, 1, 0], [1, 0, 0], [0, 0, 0]], [[0, 1, 0], [1, 1, 0], [1, 0, 0]]]
    k02 = load_model(str(tmp_path / "kite02" / "m.obj"))["verts"]
    assert k02[0].tolist() == [[1, 0, 0], [2, -1, 0], [0, 0, 0]] and k02[1].tolist() == [[2, 1, 0], [1, 0, 0], [0, 0, 0]]
    path = write_model(str(tmp_path / "zero"), "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", "newmtl x\n")
    with pytest.raises(RuntimeError, match="TinyObjReader"):
        ref_io.obj_load(path, "white")
    with pytest.raises(RuntimeError, match="zero"):
        load_model(path)


@live
def test_reference_mtl_assets(tmp_path):
    """The two MTL files the reference ships (its OBJ files are git-lfs pointers here): cube.mtl names cash_money.png, which is there."""
    if not os.path.exists(ASSETS + "/cube.mtl"):
        pytest.skip("reference assets not present on this machine")
    import shutil
    d = tmp_path / "cube"
    os.makedirs(d)
    shutil.copy(ASSETS + "/cube.mtl", d / "cube.mtl")
    shutil.copy(ASSETS + "/cash_money.png", d / "cash_money.png")
    (d / "cube.obj").write_text("mtllib cube.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                                "usemtl Default_OBJ\nf 1/1 2/2 3/3 4/4\nusemtl None\nf 1 2 3\n")
    r = ref_io.obj_load(d / "cube.obj", "white")
    assert r["materials"] == [("Default_OBJ", "cash_money.png"), ("None", "")] and r["tex_names"] == ["cash_money.png", "white"]
    s = Scene()
    s.AddMaterial("white", material(color=(1, 1, 1)))
    assert s.LoadModel(d / "cube.obj", "white") == 3
    ref_tex, c = ref_io.load_image_f(ASSETS + "/cash_money.png")
    assert c == 4 and np.array_equal(bits(np.array(s.texture_array())[:, :3]), bits(ref_tex.reshape(-1, 3)))
    prims = np.array(_view(s._lib.rth_primitives, s._h, W.Primitive))
    mats = s.material_array()
    assert int(mats[prims["matIdx"][0]]["texW"]) == ref_tex.shape[1] and int(mats[prims["matIdx"][2]]["texIdx"]) == -1


# ------------------------------------------------------------------------------------------------ goldens (run anywhere)
def test_golden_files_reproduce_the_reference_readers(tmp_path):
    """tests/golden/assets_ref.npz: input files as bytes + what the compiled reference readers returned for them."""
    g = np.load(GOLDEN)
    names = sorted({k.split("|")[1] for k in g.files if k.startswith("img|")})
    assert len(names) >= 30
    for name in names:
        f = tmp_path / name
        f.write_bytes(g[f"img|{name}|file"].tobytes())
        ours = load_texture(f)
        want, c = g[f"img|{name}|texels"], int(g[f"img|{name}|channels"])
        if c >= 3:
            assert ours.shape == want.shape and np.array_equal(bits(ours), bits(want)), name
        else:                                                # stored: stb's 8-bit grey levels
            for k in range(3):
                assert np.array_equal(bits(ours[..., k]), bits(_stb_float(want))), name
    models = sorted({k.split("|")[1] for k in g.files if k.startswith("obj|")})
    assert len(models) >= 20
    for name in models:
        path = write_model(str(tmp_path / name), g[f"obj|{name}|obj"].tobytes().decode("latin-1"), g[f"obj|{name}|mtl"].tobytes().decode("latin-1"))
        pos, force = tuple(float(x) for x in g[f"obj|{name}|pos"]), bool(g[f"obj|{name}|force"])
        ref = dict(verts=g[f"obj|{name}|verts"], uvs=g[f"obj|{name}|uvs"], tex=g[f"obj|{name}|tex"],
                   tex_names=[str(x) for x in g[f"obj|{name}|tex_names"]], diffuse=[str(x) for x in g[f"obj|{name}|diffuse"]])
        check_model(load_model(path, pos, force), ref, name)
