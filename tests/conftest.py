import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
SCENES = os.path.join(ROOT, "tests", "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"  # exists only in the build container, never on the GPU box


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mi():
    return importlib.import_module("physics-based-ray-tracing_amd")


@pytest.fixture(scope="session")
def capi():
    return importlib.import_module("physics-based-ray-tracing_amd._capi")


@pytest.fixture(scope="session")
def ob():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def known():
    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        return json.load(f)


def scene_path(name):
    return os.path.join(SCENES, name)


def oracle_render(ob, scene, seed, spp, crop=None, sample_offset=0, raw=False, n_threads=8, accel=None, flags=0):
    integ, sens = scene.integrator(), scene.sensors()[0]
    fd = integ._film_desc(scene, sens, seed, spp, crop, sample_offset, raw, flags=flags)
    osc = ob.OracleScene.from_scene(scene, accel)
    img = osc.render(sens.camera(), fd, n_threads=n_threads)
    return img, osc.last_stats
