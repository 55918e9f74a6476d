"""OBJ / PLY readers for the mesh assets the reference ships (scenes/meshes/cbox_*.obj,
scenes/meshes/teapot.ply, TestRing/TestRing.obj; binary little-endian PLY for bunny/suzanne).
Faces with more than three vertices are fan-triangulated (1,2,3),(1,3,4),... like Mitsuba's OBJ
loader.  Only positions are kept: the engine shades with face normals (out of scope:
interpolated vertex normals)."""
from __future__ import annotations

import os
import struct

import numpy as np


def load_obj(path: str):
    """-> (vertices float64 [nv,3], triangles int64 [nt,3])"""
    verts, tris = [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            if not line or line[0] == "#":
                continue
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v" and len(parts) >= 4:
                verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
            elif parts[0] == "f" and len(parts) >= 4:
                idx = []
                for tok in parts[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    tris.append((idx[0], idx[k], idx[k + 1]))
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    t = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    if len(t) and (t.min() < 0 or t.max() >= len(v)):
        raise ValueError(f"{path}: face index out of range")
    return v, t


_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h",
              "ushort": "H", "uint16": "H", "int": "i", "int32": "i", "uint": "I", "uint32": "I",
              "float": "f", "float32": "f", "double": "d", "float64": "d"}


def load_ply(path: str):
    """ASCII and binary_little_endian PLY -> (vertices [nv,3], triangles [nt,3])."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        raise ValueError(f"{path}: not a PLY file")
    header_end = data.find(b"\n", end) + 1
    header = data[:header_end].decode("ascii", errors="replace").splitlines()
    fmt = None
    elements = []  # (name, count, [(kind, ...)])
    for line in header:
        p = line.split()
        if not p:
            continue
        if p[0] == "format":
            fmt = p[1]
        elif p[0] == "element":
            elements.append([p[1], int(p[2]), []])
        elif p[0] == "property":
            if p[1] == "list":
                elements[-1][2].append(("list", p[2], p[3], p[4]))
            else:
                elements[-1][2].append(("scalar", p[1], p[2]))
    if fmt not in ("ascii", "binary_little_endian"):
        raise ValueError(f"{path}: unsupported PLY format {fmt}")
    verts, tris = None, []
    body = data[header_end:]
    if fmt == "ascii":
        tokens = body.split()
        pos = 0
        for name, count, props in elements:
            if name == "vertex":
                names = [p[2] for p in props]
                ncol = len(props)
                arr = np.array(tokens[pos:pos + count * ncol], dtype=np.float64).reshape(count, ncol)
                pos += count * ncol
                verts = arr[:, [names.index("x"), names.index("y"), names.index("z")]]
            elif name == "face":
                for _ in range(count):
                    for pr in props:
                        if pr[0] == "list":
                            n = int(tokens[pos])
                            idx = [int(t) for t in tokens[pos + 1:pos + 1 + n]]
                            pos += 1 + n
                            if pr[3] in ("vertex_indices", "vertex_index"):
                                for k in range(1, n - 1):
                                    tris.append((idx[0], idx[k], idx[k + 1]))
                        else:
                            pos += 1
            else:
                for _ in range(count):
                    for pr in props:
                        if pr[0] == "list":
                            pos += 1 + int(tokens[pos])
                        else:
                            pos += 1
    else:
        off = 0
        for name, count, props in elements:
            if name == "vertex" and all(p[0] == "scalar" for p in props):
                dt = np.dtype([(p[2], "<" + _PLY_TYPES[p[1]]) for p in props])
                arr = np.frombuffer(body, dtype=dt, count=count, offset=off)
                off += count * dt.itemsize
                verts = np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float64)
            else:
                for _ in range(count):
                    for pr in props:
                        if pr[0] == "list":
                            cf, itf = _PLY_TYPES[pr[1]], _PLY_TYPES[pr[2]]
                            (n,) = struct.unpack_from("<" + cf, body, off)
                            off += struct.calcsize(cf)
                            idx = struct.unpack_from("<%d%s" % (n, itf), body, off)
                            off += n * struct.calcsize(itf)
                            if name == "face" and pr[3] in ("vertex_indices", "vertex_index"):
                                for k in range(1, n - 1):
                                    tris.append((idx[0], idx[k], idx[k + 1]))
                        else:
                            off += struct.calcsize(_PLY_TYPES[pr[1]])
    if verts is None:
        raise ValueError(f"{path}: no vertex element")
    t = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    if len(t) and (t.min() < 0 or t.max() >= len(verts)):
        raise ValueError(f"{path}: face index out of range")
    return np.ascontiguousarray(verts, dtype=np.float64), t


def load_mesh(path: str):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        return load_obj(path)
    if ext == ".ply":
        return load_ply(path)
    raise ValueError(f"unsupported mesh format: {path}")
