"""Writes the golden images of tests/golden/*.npy with the CPU oracle (the reference itself cannot run:
Mitsuba / Dr.Jit are absent).  They pin the oracle AND the HIP path against accidental drift."""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
mi = importlib.import_module("physics-based-ray-tracing_amd")
from oracle import binding as ob  # noqa: E402
from conftest import oracle_render, scene_path  # noqa: E402

sc = mi.load_file(scene_path("cbox.xml"), res=32, spp=8)
np.save(os.path.join(HERE, "cbox_32x32_spp8_seed0.npy"), oracle_render(ob, sc, 0, 8)[0])
sc = mi.load_file(scene_path("simple.xml"), res=64, spp=4)
np.save(os.path.join(HERE, "simple_64x64_spp4_seed0.npy"), oracle_render(ob, sc, 0, 4)[0])
sc = mi.load_file(scene_path("cone_room.xml"), res=32, spp=8)      # analytic cones in radiance mode (diffuse, glass)
np.save(os.path.join(HERE, "cone_room_32x32_spp8_seed0.npy"), oracle_render(ob, sc, 0, 8)[0])
us = mi.load_file(scene_path("us_plate.xml"))
ui = us.integrator()
buf, tx = ob.OracleScene.from_scene(us).us_acquire(ui.us_params(us), 0, 32)
nz = np.argwhere(buf != 0)
np.savez_compressed(os.path.join(HERE, "us_plate_ppr32_seed0.npz"), index=nz.astype(np.int32), value=buf[buf != 0], tx=tx)
print("golden images written")
us = mi.load_file(scene_path("us_cone_box.xml"))          # analytic cone phantom
ui = us.integrator()
buf, tx = ob.OracleScene.from_scene(us).us_acquire(ui.us_params(us), 0, 8)
nz = np.argwhere(buf != 0)
np.savez_compressed(os.path.join(HERE, "us_cone_box_ppr8_seed0.npz"), index=nz.astype(np.int32), value=buf[buf != 0], tx=tx)
print("cone phantom written:", len(nz), "non-zero bins")
