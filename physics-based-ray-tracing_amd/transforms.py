"""ScalarTransform4f / Properties: the small part of Mitsuba's object model the reference's
driver touches (USMain.py:53-57,69-71,81-83 for transforms; CustomIntegrator.py:13-42,
CustomBSDF.py:8-18, CustomEmmitter.py:6-28 for props.get / has_property / [] / id)."""
from __future__ import annotations

import math

import numpy as np


class ScalarTransform4f:
    """4x4 affine transform, float64 on the host (rounded to f32 when handed to the device).
    `A @ B` composes (B applied first), exactly like mi.ScalarTransform4f."""

    __slots__ = ("matrix",)

    def __init__(self, matrix=None):
        if matrix is None:
            self.matrix = np.eye(4)
        elif isinstance(matrix, ScalarTransform4f):
            self.matrix = matrix.matrix.copy()
        else:
            m = np.asarray(matrix, dtype=np.float64)
            if m.shape == (3, 4):
                m = np.vstack([m, [0, 0, 0, 1]])
            if m.shape != (4, 4):
                raise ValueError("ScalarTransform4f expects a 4x4 matrix")
            self.matrix = m.copy()

    # -- constructors usable both as mi.ScalarTransform4f.translate(v) and ().translate(v)
    def translate(self, v):
        m = np.eye(4)
        m[:3, 3] = np.asarray(v, dtype=np.float64).reshape(3)
        return ScalarTransform4f(self.matrix @ m)

    def scale(self, v):
        v = np.asarray(v, dtype=np.float64)
        if v.ndim == 0:
            v = np.repeat(v, 3)
        m = np.diag([v[0], v[1], v[2], 1.0])
        return ScalarTransform4f(self.matrix @ m)

    def rotate(self, axis, angle):
        """Rotation by `angle` degrees about `axis` (Mitsuba convention)."""
        a = np.asarray(axis, dtype=np.float64).reshape(3)
        a = a / np.linalg.norm(a)
        t = math.radians(float(angle))
        c, s = math.cos(t), math.sin(t)
        x, y, z = a
        r = np.array([
            [c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0],
            [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0],
            [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0],
            [0, 0, 0, 1]])
        return ScalarTransform4f(self.matrix @ r)

    def look_at(self, origin, target, up):
        """Mitsuba look_at: camera space looks down +z; columns are (left, up', dir, origin)."""
        o = np.asarray(origin, dtype=np.float64).reshape(3)
        t = np.asarray(target, dtype=np.float64).reshape(3)
        u = np.asarray(up, dtype=np.float64).reshape(3)
        d = t - o
        d = d / np.linalg.norm(d)
        left = np.cross(u, d)
        n = np.linalg.norm(left)
        if n == 0:
            raise ValueError("look_at: up is parallel to the viewing direction")
        left = left / n
        newup = np.cross(d, left)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, newup, d, o
        return ScalarTransform4f(self.matrix @ m)

    def __matmul__(self, other):
        if isinstance(other, ScalarTransform4f):
            return ScalarTransform4f(self.matrix @ other.matrix)
        v = np.asarray(other, dtype=np.float64)
        if v.shape[-1] == 3:  # treated as a point
            return self.transform_affine(v)
        return self.matrix @ v

    def transform_affine(self, p):
        p = np.asarray(p, dtype=np.float64)
        return p @ self.matrix[:3, :3].T + self.matrix[:3, 3]

    def transform_vector(self, v):
        return np.asarray(v, dtype=np.float64) @ self.matrix[:3, :3].T

    def inverse(self):
        return ScalarTransform4f(np.linalg.inv(self.matrix))

    def translation(self):
        return self.matrix[:3, 3].copy()

    def __repr__(self):
        return f"ScalarTransform4f(\n{self.matrix}\n)"

    def __eq__(self, other):
        return isinstance(other, ScalarTransform4f) and np.array_equal(self.matrix, other.matrix)


Transform4f = ScalarTransform4f


class Properties:
    """Plugin constructor argument: props.get(name, default), props.has_property(name),
    props[name], props.id(), props.plugin_name()."""

    def __init__(self, plugin_name: str = "", values: dict | None = None, id_: str = ""):
        self._plugin = plugin_name
        self._values = dict(values or {})
        self._id = id_
        self._queried = set()

    def get(self, name, default=None):
        self._queried.add(name)
        return self._values.get(name, default)

    def has_property(self, name):
        return name in self._values

    def __contains__(self, name):
        return name in self._values

    def __getitem__(self, name):
        self._queried.add(name)
        if name not in self._values:
            raise KeyError(f'Property "{name}" has not been specified!')
        return self._values[name]

    def __setitem__(self, name, value):
        self._values[name] = value

    def id(self):
        return self._id

    def set_id(self, v):
        self._id = v

    def plugin_name(self):
        return self._plugin

    def property_names(self):
        return list(self._values.keys())

    def unqueried(self):
        return [k for k in self._values if k not in self._queried]

    def __repr__(self):
        return f"Properties[{self._plugin}, id={self._id!r}, {self._values}]"
