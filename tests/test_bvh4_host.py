"""Host-side check of the BVH4 the library builds (csrc/bvh_build.h: SAH BVH2 collapsed to four children per node, child boxes
rounded outward onto the node's 8-bit grid) and of the traversal scheme of csrc/device_scene.h (slots sorted by entry distance,
one stack entry per node, LDS rows + overflow), restated in plain C++ in tests/native/bvh4_check.cpp.  No GPU: the builder is
host code, and the scheme is checked against a loop over all triangles -- same closest primitive, same t, bounded visits."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SCENES

SRC = os.path.join(ROOT, "tests", "native", "bvh4_check.cpp")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bvh4") / "bvh4_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, SRC])
    return exe


def run(checker, tris, rows, rays=30000, ctrav=None):
    text = "\n".join(" ".join(repr(float(x)) for x in t.reshape(-1)) for t in tris)
    p = subprocess.run([checker, str(rows), str(rays)] + ([str(ctrav)] if ctrav is not None else []), input=text, capture_output=True,
                       text=True, timeout=300)
    rep = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0, rep
    return rep


def mesh_tris(mi, name):
    meshio = __import__("importlib").import_module(mi.__name__ + ".meshio")
    v, t = meshio.load_mesh(os.path.join(SCENES, "meshes", name))
    return v[t].astype(np.float32)


@pytest.mark.parametrize("rows", [2, 4])  # LDS rows of the stack: 2 forces the overflow path
def test_testring_bvh4_finds_what_brute_force_finds(mi, checker, rows):
    rep = run(checker, mesh_tris(mi, "TestRing.obj"), rows)
    assert rep["prims"] == 1152 and rep["outside_grid"] == 0 and rep["mismatches"] == 0 and rep["runaway"] == 0
    assert rep["max_sp"] <= rep["depth4"] <= 8            # one stack entry per level at most
    assert rep["image_bytes"] <= 80 * 1024 - 12 * 1024     # two workgroups with their stacks share the 160 KB of a CU


def test_teapot_and_a_triangle_soup(mi, checker):
    rep = run(checker, mesh_tris(mi, "teapot.ply"), 3)
    assert rep["prims"] == 2256 and rep["outside_grid"] == 0 and rep["mismatches"] == 0 and rep["runaway"] == 0
    rng = np.random.default_rng(5)
    c = rng.uniform(-1, 1, (3000, 1, 3))
    soup = (c + rng.normal(0, 0.08, (3000, 3, 3))).astype(np.float32)
    soup[::50] *= 1e-3  # tiny triangles beside large ones: coarse grids at the top, fine ones below
    rep = run(checker, soup, 3)
    assert rep["outside_grid"] == 0 and rep["mismatches"] == 0 and rep["runaway"] == 0 and rep["max_sp"] <= rep["depth4"]


def test_trees_of_the_global_memory_constant(mi, checker):
    """meshes whose leaf records exceed the LDS are built with the SAH traversal cost 1.0 (bvh_build.h BVH_CTRAV_GLOBAL): a shallower
    tree with fuller leaves, same closest hits; the stack still needs one entry per level at most"""
    tris = mesh_tris(mi, "teapot.ply")
    lds, glob = run(checker, tris, 3), run(checker, tris, 3, ctrav=1.0)
    for rep in (lds, glob):
        assert rep["outside_grid"] == 0 and rep["mismatches"] == 0 and rep["runaway"] == 0 and rep["max_sp"] <= rep["depth4"] <= 32
    assert glob["nodes4"] <= lds["nodes4"]     # a dearer node step never buys more nodes
