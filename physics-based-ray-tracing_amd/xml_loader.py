"""Mitsuba-3 XML scene description -> the nested-dict form mi.load_dict() takes.

Covers what the reference's scene files use (scenes/cbox.xml, scenes/simple.xml,
MitsubaScenes/*.xml): <default>/$name substitution (overridable as load_file(path, spp=..)),
<ref id>, <transform> with lookat/translate/rotate/scale/matrix applied in listed order (each new
operation left-multiplies), scalar/rgb/point/vector/string/boolean properties
and the non-standard <float_array> of MitsubaScenes/Sphere_Box.xml:14."""
from __future__ import annotations

import os
import re
import xml.etree.ElementTree as ET

import numpy as np

from .transforms import ScalarTransform4f

PLUGIN_TAGS = {"integrator", "sensor", "bsdf", "shape", "emitter", "sampler", "film", "rfilter", "texture",
               "medium", "phase", "volume"}
# default key of an anonymous nested plugin
_NESTED_KEY = {"bsdf": "bsdf", "emitter": "emitter", "sampler": "sampler", "film": "film", "rfilter": "rfilter",
               "sensor": "sensor", "integrator": "integrator"}
_CAMEL = re.compile(r"(?<!^)(?=[A-Z])")


def _floats(s: str):
    return [float(t) for t in re.split(r"[\s,]+", s.strip()) if t]


def _subst(value: str, env: dict) -> str:
    def rep(m):
        k = m.group(1)
        if k not in env:
            raise KeyError(f'undefined XML parameter "${k}"')
        return str(env[k])

    return re.sub(r"\$(\w+)", rep, value)


def _vec3(el, env, default=None):
    if "value" in el.attrib:
        v = _floats(_subst(el.attrib["value"], env))
        if len(v) == 1:
            v = v * 3
        return v
    d = default if default is not None else [0.0, 0.0, 0.0]
    return [float(_subst(el.attrib.get(a, str(d[i])), env)) for i, a in enumerate("xyz")]


def _transform(el, env, order="listed") -> ScalarTransform4f:
    # Mitsuba: listed order, every new operation left-multiplies (M = op_n ... op_2 op_1).  order="intent" (the
    # transform_order keyword of load_file: a loader option, NOT one of the $substitutions of `env`) composes the other way
    # round, M = op_1 op_2 ... op_n: the reading under which the phantoms of MitsubaScenes/*.xml, which list translate,
    # rotate, scale, describe what USMain.py:69-71 builds (T @ R @ S).
    intent = order == "intent"
    m = np.eye(4)
    for op in el:
        a = {k: _subst(v, env) for k, v in op.attrib.items()}
        if op.tag == "lookat":
            t = ScalarTransform4f().look_at(_floats(a["origin"]), _floats(a["target"]), _floats(a.get("up", "0,1,0")))
        elif op.tag == "translate":
            t = ScalarTransform4f().translate(_vec3(op, env))
        elif op.tag == "scale":
            t = ScalarTransform4f().scale(_vec3(op, env, [1.0, 1.0, 1.0]))
        elif op.tag == "rotate":
            if "axis" in a:
                axis = _floats(a["axis"])
            elif "value" in a:
                axis = _floats(a["value"])
            else:
                axis = [float(a.get(k, 0.0)) for k in "xyz"]
            t = ScalarTransform4f().rotate(axis, float(a["angle"]))
        elif op.tag == "matrix":
            v = _floats(a["value"])
            t = ScalarTransform4f(np.asarray(v, dtype=np.float64).reshape(4, 4) if len(v) == 16 else
                                  np.vstack([np.hstack([np.asarray(v).reshape(3, 3), np.zeros((3, 1))]), [0, 0, 0, 1]]))
        else:
            raise ValueError(f"unsupported transform operation <{op.tag}>")
        m = (m @ t.matrix) if intent else (t.matrix @ m)
    return ScalarTransform4f(m)


def _snake(name: str) -> str:
    return _CAMEL.sub("_", name).lower() if any(c.isupper() for c in name) else name


def _plugin(el, env, base_dir, counters, order="listed") -> dict:
    d = {"type": _subst(el.attrib["type"], env)} if el.tag != "scene" else {"type": "scene"}
    if "id" in el.attrib:
        d["id"] = el.attrib["id"]
    for ch in el:
        tag = ch.tag
        name = _snake(ch.attrib.get("name", ""))
        if tag == "default":
            continue
        if tag in PLUGIN_TAGS:
            key = name or ch.attrib.get("id") or _NESTED_KEY.get(tag)
            if not key or key in d:
                counters[tag] = counters.get(tag, 0) + 1
                key = f"{tag}_{counters[tag]}"
            d[key] = _plugin(ch, env, base_dir, counters, order)
        elif tag == "ref":
            counters["ref"] = counters.get("ref", 0) + 1
            d[name or f"ref_{counters['ref']}"] = {"type": "ref", "id": ch.attrib["id"]}
        elif tag in ("float",):
            v = _subst(ch.attrib["value"], env)
            d[name] = float(v)
        elif tag == "integer":
            d[name] = int(float(_subst(ch.attrib["value"], env)))
        elif tag == "boolean":
            d[name] = _subst(ch.attrib["value"], env).strip().lower() == "true"
        elif tag == "string":
            v = _subst(ch.attrib["value"], env)
            if name == "filename" and not os.path.isabs(v):
                v = os.path.join(base_dir, v)
            d[name] = v
        elif tag in ("rgb", "spectrum", "color"):
            v = _floats(_subst(ch.attrib["value"], env))
            d[name] = {"type": "rgb", "value": v * 3 if len(v) == 1 else v}
        elif tag in ("point", "vector"):
            d[name] = _vec3(ch, env)
        elif tag == "transform":
            d[name] = _transform(ch, env, order)
        elif tag == "float_array":
            d[name] = np.asarray(_floats(_subst(ch.attrib["value"], env)), dtype=np.float32)
        elif tag == "include":
            sub = load_xml_to_dict(os.path.join(base_dir, _subst(ch.attrib["filename"], env)), transform_order=order, **env)
            for k, v in sub.items():
                if k != "type":
                    d[k] = v
        else:
            raise ValueError(f"unsupported XML element <{tag}>")
    return d


def load_xml_to_dict(path: str, transform_order: str = "listed", **overrides) -> dict:
    """overrides: values for the scene's <default name=...> substitutions (mi.load_file(path, res=512, spp=256)).
    transform_order: "listed" (Mitsuba) or "intent" (DESIGN D14) -- a loader option, kept apart from the substitutions so
    that neither a <default name="transform_order"> in a scene file nor a caller's substitution of that name can flip it."""
    if transform_order not in ("listed", "intent"):
        raise ValueError(f"transform_order must be 'listed' or 'intent', not {transform_order!r}")
    tree = ET.parse(path)
    root = tree.getroot()
    if root.tag != "scene":
        raise ValueError(f"{path}: root element must be <scene>")
    env = {}
    for el in root.iter("default"):
        env[el.attrib["name"]] = el.attrib["value"]
    for k, v in overrides.items():
        env[k] = v
    return _plugin(root, env, os.path.dirname(os.path.abspath(path)), {}, transform_order)
