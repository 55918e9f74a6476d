"""OBJ / PLY readers for the mesh assets the reference ships (scenes/meshes/cbox_*.obj,
scenes/meshes/teapot.ply, TestRing/TestRing.obj; binary little-endian PLY for bunny/suzanne).
Faces with more than three vertices are fan-triangulated (1,2,3),(1,3,4),... like Mitsuba's OBJ
loader.  Positions, and -- with normals=True -- the vertex normals Mitsuba interpolates into the shading normal
(OBJ `vn` through the faces' v//vn or v/vt/vn indices, PLY nx ny nz); texture coordinates are not read."""
from __future__ import annotations

import os
import struct

import numpy as np


def load_obj(path: str, normals: bool = False):
    """-> (vertices float64 [nv,3], triangles int64 [nt,3]); with normals=True also the per-triangle vertex normals
    float64 [nt,3,3] (None if the file has no `vn` or a face lacks normal indices)"""
    verts, tris, vns, ntris = [], [], [], []
    all_have_n = True
    with open(path, "r", errors="replace") as f:
        for line in f:
            if not line or line[0] == "#":
                continue
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v" and len(parts) >= 4:
                verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
            elif parts[0] == "vn" and len(parts) >= 4:
                vns.append((float(parts[1]), float(parts[2]), float(parts[3])))
            elif parts[0] == "f" and len(parts) >= 4:
                idx, nidx = [], []
                for tok in parts[1:]:
                    fields = tok.split("/")
                    i = int(fields[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                    if len(fields) >= 3 and fields[2]:
                        j = int(fields[2])
                        nidx.append(j - 1 if j > 0 else len(vns) + j)
                    else:
                        all_have_n = False
                        nidx.append(-1)
                for k in range(1, len(idx) - 1):
                    tris.append((idx[0], idx[k], idx[k + 1]))
                    ntris.append((nidx[0], nidx[k], nidx[k + 1]))
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    t = np.asarray(tris, dtype=np.int64).reshape(-1, 3)
    if len(t) and (t.min() < 0 or t.max() >= len(v)):
        raise ValueError(f"{path}: face index out of range")
    if not normals:
        return v, t
    tn = None
    if vns and all_have_n and len(t):
        vn = np.asarray(vns, dtype=np.float64).reshape(-1, 3)
        nt = np.asarray(ntris, dtype=np.int64).reshape(-1, 3)
        if nt.min() < 0 or nt.max() >= len(vn):
            raise ValueError(f"{path}: normal index out of range")
        tn = vn[nt]
    return v, t, tn


_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h",
              "ushort": "H", "uint16": "H", "int": "i", "int32": "i", "uint": "I", "uint32": "I",
              "float": "f", "float32": "f", "double": "d", "float64": "d"}


def load_ply(path: str, normals: bool = False):
    """ASCII and binary_little_endian PLY -> (vertices [nv,3], triangles [nt,3]); with normals=True also the per-triangle
    vertex normals [nt,3,3] (None without nx / ny / nz vertex properties)."""
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
    verts, tris, vnorm = None, [], None
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
                if all(c in names for c in ("nx", "ny", "nz")):
                    vnorm = arr[:, [names.index("nx"), names.index("ny"), names.index("nz")]]
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
                if all(c in dt.names for c in ("nx", "ny", "nz")):
                    vnorm = np.stack([arr["nx"], arr["ny"], arr["nz"]], axis=1).astype(np.float64)
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
    v = np.ascontiguousarray(verts, dtype=np.float64)
    if not normals:
        return v, t
    return v, t, (np.asarray(vnorm, dtype=np.float64)[t] if vnorm is not None and len(t) else None)


def load_mesh(path: str, normals: bool = False):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        return load_obj(path, normals)
    if ext == ".ply":
        return load_ply(path, normals)
    raise ValueError(f"unsupported mesh format: {path}")
