#!/usr/bin/env python3
"""The reference's ultrasound driver flow on this library: scene dict -> acquisition -> delay-and-sum -> envelope ->
log compression -> finite-difference roughness loop (what USMain.py does at :26-90, :93-224, :257-289), without the
plotting.  Writes the B-mode image and the channel buffer as .npy.   python examples/us_bmode.py [out_dir]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import pbrt_amd as mi                       # was: import mitsuba as mi
import pbrt_amd.drjit_compat as dr          # was: import drjit as dr

mi.set_variant("llvm_ad_mono")              # accepted; the one backend is HIP on gfx950
out_dir = sys.argv[1] if len(sys.argv) > 1 else "."
T = mi.ScalarTransform4f

scene = mi.load_dict({
    "type": "scene",
    "integrator": {"type": "ultrasound_integrator", "max_depth": 10, "sampling_rate": 50e6, "frequency": 5e6,
                   "sound_speed": 1540, "attenuation": 0.2, "wave_cycles": 5, "main_beam_angle": 24, "cutoff_angle": 30,
                   "n_elements": 64, "pitch": 1.2e-4, "time_samples": 10000, "angles": dr.linspace(mi.Float, -15, 15, 5),
                   "paths_per_ray": 4096, "seed": 1},
    "sensor": {"type": "ultrasound_sensor", "num_elements_lateral": 1280, "elements_width": 0.003, "elements_height": 0.01,
               "pitch": 0.0003, "center_frequency": 5e6, "sound_speed": 1540, "directivity": 1.0,
               "to_world": T().look_at(origin=[0, 0, 0], target=[0, 0, 0.03], up=[0, 1, 0])},
    "flat_plate": {"type": "rectangle",
                   "to_world": T().translate([0, 0, 0.05]) @ T().rotate([0, 1, 0], 45) @ T().scale([0.17, 0.17, 0.14]),
                   "bsdf": {"type": "ultrasound_bsdf", "impedance": 7.8, "roughness": 0.7}},
    "wall_back": {"type": "rectangle",
                  "to_world": T().translate([0, 0, 1]) @ T().rotate([0, 1, 0], 180) @ T().scale([0.05, 0.05, 1]),
                  "bsdf": {"type": "ultrasound_bsdf", "impedance": 7.8, "roughness": 0.7}},
})

t = time.perf_counter()
display, bmode, (x_scan, z_scan) = mi.us_render(scene, x_range=(-0.02, 0.02), z_range=(0.02, 0.08))
print(f"B-mode {display.shape[0]} x {display.shape[1]} pixels in {(time.perf_counter() - t) * 1e3:.1f} ms; "
      f"channel_buf sum {float(np.sum(scene.integrator().channel_buf)):.4g}, max {float(np.max(scene.integrator().channel_buf)):.4g}")
np.save(os.path.join(out_dir, "bmode_display.npy"), display)
np.save(os.path.join(out_dir, "channel_buf.npy"), np.asarray(scene.integrator().channel_buf))

# the finite-difference loop of USMain.py:257-289; every forward run uses the same seed (common random numbers)
params = mi.traverse(scene)
key = [k for k in params.keys() if k.endswith("flat_plate.bsdf.roughness")][0]
target = bmode.astype(np.float64)


def forward(rough):
    params[key] = rough
    params.update()
    return mi.us_render(scene, x_range=(-0.02, 0.02), z_range=(0.02, 0.08))[1].astype(np.float64)


rough, eps = 0.5, 1e-2
scale = float(np.mean(target ** 2))
for it in range(5):
    f0 = float(np.mean((forward(rough) - target) ** 2)) / scale
    f1 = float(np.mean((forward(rough + eps) - target) ** 2)) / scale
    grad = (f1 - f0) / eps
    rough = float(np.clip(rough - 0.05 * np.sign(grad), 1e-4, 1.0))
    print(f"iter {it}: relative loss {f0:.4g}, d loss / d roughness {grad:.4g}, roughness -> {rough:.3f}")
