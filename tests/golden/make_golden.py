"""Generates tests/golden/known_answers.json (run in the build container, where /root/reference exists;
the JSON is committed, this script documents how it was made).

K1 comes from IMPORTING the reference's only importable file, sampling_test.py (pure NumPy).
K2-K6 evaluate, in float64 NumPy, the closed-form expressions of the reference lines cited next to
each entry (no Mitsuba needed).  K7 is the world-space geometry table of SURVEY.md App. E.
"""
import contextlib
import importlib.util
import io
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_sampling_test():
    spec = importlib.util.spec_from_file_location("ref_sampling_test", os.path.join(REF, "sampling_test.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    out = {}
    st = load_sampling_test()
    # ---- K1: GGX (sampling_test.py:3-23 inverse CDF, :25-43 D*sin normalised by its max)
    theta = np.linspace(0, 90, 100)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):  # ggx_pdf prints np.max(pdf)
        pdf_norm = st.ggx_pdf(theta, 0.5)
    pdf_max = float(buf.getvalue().strip())
    np.random.seed(0)
    seeded = st.sample_ggx_scattering_angle(0.5, 5)
    xi = np.array([0.0, 0.25, 0.5, 0.75])
    inv_cdf = np.arccos(np.sqrt((1 - xi) / (1 + (0.5 ** 2 - 1) * xi))) * 180 / np.pi  # sampling_test.py:18-21
    out["K1_ggx"] = {
        "alpha": 0.5,
        "pdf_max": pdf_max,                        # == literal at CustomBSDF.py:81
        "pdf_argmax_deg": float(theta[int(np.argmax(pdf_norm))]),
        "theta_grid_deg": theta.tolist(),
        "pdf_normalised": pdf_norm.tolist(),
        "seed0_samples_deg": seeded.tolist(),
        "xi": xi.tolist(),
        "inverse_cdf_deg": inv_cdf.tolist(),
    }
    # ---- K2: transmission delays (CustomIntegrator.py:246-257), Sphere_Box parameters
    N, pitch, c = 64, 1.2e-4, 1480.0
    ang = np.array([-15, -7.5, 0, 7.5, 15], dtype=np.float64)
    elem_x = (pitch * (np.arange(N, dtype=np.float32) - (N - 1) / 2)).astype(np.float32)       # :248
    ag, eg = np.meshgrid(np.deg2rad(ang), elem_x, indexing="ij")                               # :251
    tx = ((eg * np.sin(ag)) / c).astype(np.float32)                                            # :254,257
    out["K2_tx_delay"] = {"n_elements": N, "pitch": pitch, "sound_speed": c, "angles_deg": ang.tolist(),
                          "elem_x": elem_x.astype(np.float64).tolist(), "tx_delay": tx.astype(np.float64).tolist()}
    # ---- K3: attenuation factor exp(-alpha f 1e-6 d / 8.686) (CustomIntegrator.py:328)
    d = np.array([0.02, 0.05, 0.10])
    out["K3_attenuation"] = {"attenuation": 0.1, "frequency": 3e6, "distance": d.tolist(),
                             "factor": np.exp(-0.1 * 3e6 * 1e-6 * d / 8.686).tolist()}
    # ---- K4: directivity_weight_i trapezoid (CustomIntegrator.py:289-304), alpha_m 24, alpha_c 30 deg
    a = np.array([10.0, 24.0, 27.0, 30.0, 31.0])
    am, ac = 24.0, 30.0
    w = np.where(a <= am, 1.0, np.where(a <= ac, (ac - a) / (ac - am), 0.0))
    out["K4_directivity"] = {"main_beam_angle": am, "cutoff_angle": ac, "angle_deg": a.tolist(), "weight": w.tolist()}
    # ---- K5: impedance coefficients (CustomBSDF.py:103-124,137,142,154), Z1 = 7.8, Z2 = 1.2
    Z1, Z2 = 7.8, 1.2
    rows = []
    for cosTr in (1.0, 0.995, 0.99, 0.98):
        ratio = Z1 / Z2
        sa = 1 - ratio ** 2 * (1 - cosTr ** 2)
        cosTt = math.sqrt(max(sa, 0.0))
        Ar = (Z1 * cosTr - Z2 * cosTt) / (Z1 * cosTr + Z2 * cosTt)
        rows.append({"cosTr": cosTr, "Ar": Ar, "At": 1 - Ar, "Ar2": Ar * Ar, "pdf_reflect": 1 / (4 * cosTr),
                     "tir": bool(sa < 0)})
    out["K5_impedance"] = {"Z1": Z1, "Z2": Z2, "rows": rows,
                           "tir_cos_threshold": math.sqrt(1 - (Z2 / Z1) ** 2)}
    # ---- K6: CustomSensor.put_data known answer (CustomSensor.py:29-59, commented test :80-100)
    Nn, pit, fs, T = 5, 1.0, 10.0, 20
    rays = [(-2.0, 1.0, (0, 0, -1), 1.0), (0.0, 1.5, (0, 0, -1), 2.0), (2.0, 0.5, (0, 0.8, -1), 1.0), (10.0, 1.0, (0, 0, -1), 3.0)]
    buf6 = np.zeros((Nn, T), np.float32)
    for x, t, dd, amp in rays:
        idx = int(np.round(x / pit + Nn / 2))                   # :36 (np.round: half-to-even)
        it = int(np.round(t * fs))                              # :43
        dv = -np.asarray(dd, dtype=np.float64)
        dv /= np.linalg.norm(dv)                                # :46
        gain = max(0.0, float(dv @ np.array([0, 0, 1.0])))      # :51
        if 0 <= idx < Nn and 0 <= it < T:                       # :58
            buf6[idx, it] += amp * gain                         # :59
    out["K6_put_data"] = {"number_of_elements": Nn, "pitch": pit, "sample_rate": fs, "time_samples": T,
                          "rays": [{"x": r[0], "time": r[1], "d": list(r[2]), "amplitude": r[3]} for r in rays],
                          "nonzero": [{"element": int(i), "sample": int(j), "value": float(buf6[i, j])}
                                      for i, j in np.argwhere(buf6 != 0)]}
    # ---- K7: world-space geometry
    out["K7_geometry"] = {
        "cbox": {"floor": {"y": -1, "n": [0, 1, 0]}, "ceiling": {"y": 1, "n": [0, -1, 0]}, "back": {"z": -1, "n": [0, 0, 1]},
                 "green": {"x": -1, "n": [1, 0, 0]}, "red": {"x": 1, "n": [-1, 0, 0]},
                 "luminaire": {"y": 0.99, "n": [0, -1, 0], "area": 0.25, "half": 0.25},
                 "mirror_sphere": {"c": [-0.3, -0.5, 0.2], "r": 0.5}, "glass_sphere": {"c": [0.5, -0.75, -0.2], "r": 0.25},
                 "camera_origin": [0, 0, 4], "x_fov_deg": 39.3077},
        "sphere_box_mitsuba_semantics": {"sphere": {"c": [0, 0, 0.0048], "r": 0.06}, "box_back": {"z": -0.37, "n": [0, 0, -1]},
                                         "box_left": {"x": 0.03, "n": [1, 0, 0]}, "box_right": {"x": -0.03, "n": [-1, 0, 0]},
                                         "box_top": {"y": -0.03, "n": [0, -1, 0]}, "box_bottom": {"y": 0.03, "n": [0, 1, 0]}},
        "sphere_box_intent": {"sphere": {"c": [0, 0, 0.08], "r": 0.06}, "box_back": {"z": 0.37, "n": [0, 0, -1]},
                              "box_left": {"x": -0.15, "n": [1, 0, 0]}, "box_right": {"x": 0.15, "n": [-1, 0, 0]},
                              "box_top": {"y": 0.15, "n": [0, -1, 0]}, "box_bottom": {"y": -0.15, "n": [0, 1, 0]}},
        "testring": {"n_triangles": 1152, "n_vertices": 576, "bbox_lo": [-0.06, -0.06, 0], "bbox_hi": [0.06, 0.06, 0.05]},
        "teapot": {"n_triangles": 2256, "n_vertices": 1177},
        "simple_x_fov_deg": 34.022057,
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote known_answers.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
