"""One-off conversion (run in the build container, where /root/reference exists): the two Sketchfab glTF models that ship
with the reference (CC-BY-4.0, see magr_ray_tracer_amd/assets/ATTRIBUTION.md) -> compact mesh files used by config 5."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magr_ray_tracer_amd import gltf  # noqa: E402

SRC = "/root/reference/assets"
OUT = os.path.join(ROOT, "magr_ray_tracer_amd", "assets")
for name, out in (("robo-orb", "robo_orb.npz"), ("terrarium_bot", "terrarium_bot.npz")):
    parts = gltf.load_gltf(os.path.join(SRC, name, "scene.gltf"))
    tris = sum(len(p["indices"]) for p in parts)
    verts = sum(len(p["vertices"]) for p in parts)
    gltf.pack(parts, os.path.join(OUT, out), source=name)
    print(name, "meshes", len(parts), "tris", tris, "verts", verts, "->", out, os.path.getsize(os.path.join(OUT, out)) // 1024, "KiB")
