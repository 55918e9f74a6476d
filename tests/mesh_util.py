"""test helper: writes a UV sphere as an OBJ with (optionally) exact radial vertex normals"""
import math


def write_uv_sphere_obj(path, radius=1.0, n_lat=8, n_lon=12, normals=True, center=(0.0, 0.0, 0.0)):
    verts, faces = [], []
    for i in range(n_lat + 1):
        th = math.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * math.pi * j / n_lon
            verts.append((math.sin(th) * math.cos(ph), math.sin(th) * math.sin(ph), math.cos(th)))
    idx = lambda i, j: i * n_lon + (j % n_lon) + 1
    for i in range(n_lat):
        for j in range(n_lon):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            if i > 0:
                faces.append((a, b, d))
            if i < n_lat - 1:
                faces.append((b, c, d))
    with open(path, "w") as f:
        for x, y, z in verts:
            f.write(f"v {center[0] + radius * x:.9g} {center[1] + radius * y:.9g} {center[2] + radius * z:.9g}\n")
        if normals:
            for x, y, z in verts:
                f.write(f"vn {x:.9g} {y:.9g} {z:.9g}\n")
        for a, b, c in faces:
            f.write(f"f {a}//{a} {b}//{b} {c}//{c}\n" if normals else f"f {a} {b} {c}\n")
    return len(faces)
