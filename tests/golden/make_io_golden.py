"""Generates tests/golden/assets_ref.npz by running the REFERENCE's own asset readers (oracle/_ref/libref_io.so: its vendored stb_image and
tinyobjloader headers compiled where they lie by oracle/build_ref.sh; nothing of the reference's source is stored here) on small
synthetic input files.  The fixture holds the input FILES as bytes together with what the reference readers returned:

    img|<name>|file      the image file              img|<name>|channels   stb_image's channel count
    img|<name>|texels    LoadImageF's (h, w, 3) float32 texels (>= 3 channels) or stb_image's 8-bit grey levels (1-2 channels)
    obj|<name>|obj, mtl  the model files              obj|<name>|pos, force  LoadModel's _pos / _forceDefaultMat
    obj|<name>|verts, uvs, tex, tex_names, diffuse    what Scene::LoadModel hands to AddTriangle (see tests/ref_io.py obj_load)

    python tests/golden/make_io_golden.py          # needs /root/reference (for the build) and Pillow (to write the JPEG inputs)
"""
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_io  # noqa: E402
from test_io_postproc_cpu import _png_bytes  # noqa: E402
from test_ref_io_cpu import jpeg_cases, random_obj, write_model  # noqa: E402

out = {}
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(77)


def add_image(name, data):
    path = os.path.join(tmp, name)
    with open(path, "wb") as f:
        f.write(data)
    tex, c = ref_io.load_image_f(path)
    out[f"img|{name}|file"] = np.frombuffer(data, np.uint8)
    out[f"img|{name}|channels"] = np.int32(c)
    out[f"img|{name}|texels"] = tex if c >= 3 else ref_io.load_image_u8(path)[..., 0]


for name, im, kw in jpeg_cases():
    if im.size[0] * im.size[1] > 61 * 83:
        continue
    p = os.path.join(tmp, "j.jpg")
    im.save(p, "JPEG", **kw)
    add_image(name + ".jpg", open(p, "rb").read())
for ctype, chan in ((0, 1), (2, 3), (4, 2), (6, 4)):
    for depth in (8, 16):
        add_image(f"c{ctype}_d{depth}.png", _png_bytes(rng.integers(0, 1 << depth, (9, 11, chan)), ctype, depth))
for depth in (1, 2, 4, 8):
    pix = rng.integers(0, 1 << depth, (9, 11, 1))
    add_image(f"p{depth}.png", _png_bytes(pix, 3, depth, palette=rng.integers(0, 256, (1 << depth, 3))))
    if depth < 8:
        add_image(f"g{depth}.png", _png_bytes(pix, 0, depth))
for ctype, chan, depth in ((2, 3, 8), (6, 4, 16), (0, 1, 2)):
    add_image(f"adam7_c{ctype}_d{depth}.png", _png_bytes(rng.integers(0, 1 << depth, (9, 11, chan)), ctype, depth, interlace=True))
w, h = 13, 7
for name, bpp, top in (("t24", 24, False), ("t32", 32, True)):
    pix = rng.integers(0, 256, (h, w, bpp // 8)).astype(np.uint8)
    add_image(name + ".tga", struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, bpp, (0x20 if top else 0) | (8 if bpp == 32 else 0)) + pix.tobytes())
rgbe = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
add_image("flat.hdr", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode() + rgbe.tobytes())

k = 0
while sum(1 for n in out if n.endswith("|obj")) < 24:
    obj, mtl = random_obj(rng, with_mtl=k % 5 != 0)
    pos = (0.0, 0.0, 0.0) if k % 3 else tuple(float(x) for x in rng.uniform(-5, 5, 3))
    force = k % 7 == 0
    path = write_model(os.path.join(tmp, f"m{k}"), obj, mtl)
    k += 1
    try:
        r = ref_io.obj_load(path, "white", pos, force)
    except RuntimeError:
        continue
    name = f"m{k:02d}"
    out[f"obj|{name}|obj"] = np.frombuffer(obj.encode("latin-1"), np.uint8)
    out[f"obj|{name}|mtl"] = np.frombuffer(mtl.encode("latin-1"), np.uint8)
    out[f"obj|{name}|pos"] = np.array(pos, np.float32)
    out[f"obj|{name}|force"] = np.int32(force)
    out[f"obj|{name}|verts"], out[f"obj|{name}|uvs"], out[f"obj|{name}|tex"] = r["verts"], r["uvs"], r["tex"]
    out[f"obj|{name}|tex_names"] = np.array(r["tex_names"] or [""])
    out[f"obj|{name}|diffuse"] = np.array([d for _, d in r["materials"]] or [""])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "assets_ref.npz"), **out)
print(len([n for n in out if n.endswith("|file")]), "images,", len([n for n in out if n.endswith("|obj")]), "models ->", "tests/golden/assets_ref.npz",
      os.path.getsize(os.path.join(ROOT, "tests", "golden", "assets_ref.npz")), "bytes")
