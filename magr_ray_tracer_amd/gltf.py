"""Minimal glTF 2.0 mesh reader: positions + indices + node matrices -> world-space triangle
soup per material.  Enough for static Sketchfab exports like the reference's assets/robo-orb and assets/terrarium_bot
(the reference itself expected OBJ exports of them that were never committed, src/scene.cpp:47-61)."""
import json
import os

import numpy as np

_CT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NC = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def _node_matrix(n):
    if "matrix" in n:
        return np.array(n["matrix"], dtype=np.float64).reshape(4, 4).T      # glTF stores column-major
    m = np.eye(4)
    r = _quat_to_mat(n.get("rotation", [0, 0, 0, 1]))
    s = np.array(n.get("scale", [1, 1, 1]), dtype=np.float64)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = n.get("translation", [0, 0, 0])
    return m


def load_gltf(path):
    """Returns a list of dicts {name, material, vertices (n,3) f32 world space, indices (m,3) u32}."""
    g = json.load(open(path))
    base = os.path.dirname(path)
    bufs = [np.fromfile(os.path.join(base, b["uri"]), dtype=np.uint8) for b in g["buffers"]]

    def accessor(i):
        a = g["accessors"][i]
        bv = g["bufferViews"][a["bufferView"]]
        dt, nc = np.dtype(_CT[a["componentType"]]), _NC[a["type"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or dt.itemsize * nc
        raw = bufs[bv["buffer"]]
        if stride == dt.itemsize * nc:
            return np.frombuffer(raw, dtype=dt, count=a["count"] * nc, offset=off).reshape(a["count"], nc)
        out = np.zeros((a["count"], nc), dtype=dt)
        for k in range(a["count"]):
            out[k] = np.frombuffer(raw, dtype=dt, count=nc, offset=off + k * stride)
        return out

    mats = [m.get("name", f"mat{i}") for i, m in enumerate(g.get("materials", []))]
    parts = []

    def visit(ni, parent):
        n = g["nodes"][ni]
        world = parent @ _node_matrix(n)
        if "mesh" in n:
            mesh = g["meshes"][n["mesh"]]
            for p in mesh["primitives"]:
                if p.get("mode", 4) != 4:
                    continue
                pos = accessor(p["attributes"]["POSITION"]).astype(np.float64)
                idx = accessor(p["indices"]).reshape(-1, 3).astype(np.uint32) if "indices" in p else np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
                w = (pos @ world[:3, :3].T + world[:3, 3]).astype(np.float32)
                parts.append(dict(name=mesh.get("name", ""), material=mats[p["material"]] if "material" in p else "", vertices=w, indices=idx))
        for c in n.get("children", []):
            visit(c, world)

    for root in g["scenes"][g.get("scene", 0)]["nodes"]:
        visit(root, np.eye(4))
    return parts


def pack(parts, path, **meta):
    """Compact mesh file: concatenated vertices/indices + per-triangle material ids."""
    names = sorted({p["material"] for p in parts})
    V, I, M, off = [], [], [], 0
    for p in parts:
        V.append(p["vertices"]); I.append(p["indices"] + off); M.append(np.full(len(p["indices"]), names.index(p["material"]), np.uint8))
        off += len(p["vertices"])
    np.savez_compressed(path, vertices=np.concatenate(V), indices=np.concatenate(I), material=np.concatenate(M),
                        material_names=np.array(names), **{k: np.array(v) for k, v in meta.items()})


def load_packed(path):
    g = np.load(path)
    return g["vertices"], g["indices"], g["material"], [str(s) for s in g["material_names"]]
