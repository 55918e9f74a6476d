"""The committed golden films / channel buffers (tests/golden/make_images.py) against the oracle as it is built now:
radiance bit for bit, ultrasound to the last float (same machine arithmetic, scalar libm).  The GPU suites compare
the HIP path with the same files."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, oracle_render, scene_path


@pytest.mark.parametrize("scene,kw,spp,name", [("cbox.xml", dict(res=32, spp=8), 8, "cbox_32x32_spp8_seed0.npy"),
                                               ("simple.xml", dict(res=64, spp=4), 4, "simple_64x64_spp4_seed0.npy"),
                                               ("cone_room.xml", dict(res=32, spp=8), 8, "cone_room_32x32_spp8_seed0.npy")])
def test_oracle_reproduces_golden_film(mi, ob, scene, kw, spp, name):
    g = np.load(os.path.join(GOLDEN, name))
    img, _ = oracle_render(ob, mi.load_file(scene_path(scene), **kw), 0, spp)
    assert img.shape == g.shape and np.array_equal(img, g) and g.mean() > 0
    one, _ = oracle_render(ob, mi.load_file(scene_path(scene), **kw), 0, spp, n_threads=1)   # thread count is not an input
    assert np.array_equal(one, g)


@pytest.mark.parametrize("scene,ppr,name", [("us_plate.xml", 32, "us_plate_ppr32_seed0.npz"), ("us_cone_box.xml", 8, "us_cone_box_ppr8_seed0.npz")])
def test_oracle_reproduces_golden_channel_buffer(mi, ob, scene, ppr, name):
    g = np.load(os.path.join(GOLDEN, name))
    us = mi.load_file(scene_path(scene))
    ui = us.integrator()
    buf, tx = ob.OracleScene.from_scene(us).us_acquire(ui.us_params(us), 0, ppr)
    ref = np.zeros_like(buf)
    ref[tuple(g["index"].T)] = g["value"]
    assert len(g["value"]) > 500 and np.array_equal(buf != 0, ref != 0)
    assert np.allclose(buf, ref, rtol=1e-6, atol=0) and np.array_equal(tx, g["tx"])
