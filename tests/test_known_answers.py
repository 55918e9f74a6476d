"""K1-K6: the oracle and the host logic against the known answers derived from
the reference's own formulas / its importable sampling_test.py (tests/golden/make_golden.py)."""
import ctypes as C

import numpy as np
import pytest


def test_k1_ggx_matches_reference_sampling_test(ob, known):
    k = known["K1_ggx"]
    got = ob.ggx_angle_deg(k["alpha"], np.array(k["xi"]))
    assert np.allclose(got, k["inverse_cdf_deg"], rtol=0, atol=1e-9)
    raw = ob.ggx_pdf_raw(k["alpha"], np.array(k["theta_grid_deg"]))
    assert raw.max() == pytest.approx(k["pdf_max"], rel=1e-14)          # the literal at CustomBSDF.py:81
    assert k["pdf_max"] == 0.2386683650839149
    assert np.allclose(raw / raw.max(), k["pdf_normalised"], rtol=1e-12, atol=1e-15)
    assert k["theta_grid_deg"][int(np.argmax(raw))] == pytest.approx(k["pdf_argmax_deg"])
    # the seeded draw of sampling_test.py: np.random.seed(0); uniform(0,1,5) through the inverse CDF
    np.random.seed(0)
    xi = np.random.uniform(0, 1, 5)
    assert np.allclose(ob.ggx_angle_deg(k["alpha"], xi), k["seed0_samples_deg"], rtol=0, atol=1e-9)


def _us_params(capi, k2, **kw):
    p = capi.UsParams()
    p.n_elements, p.pitch, p.sound_speed = k2["n_elements"], k2["pitch"], k2["sound_speed"]
    p.n_angles = len(k2["angles_deg"])
    for i, a in enumerate(k2["angles_deg"]):
        p.angles_deg[i] = a
    p.time_samples, p.max_depth, p.fs, p.frequency = 100, 1, 5e7, 3e6
    ident = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    for i, v in enumerate(ident):
        p.sensor_to_world[i] = v
    return p


def test_k2_tx_delays_oracle_and_library(ob, capi, known):
    k = known["K2_tx_delay"]
    p = _us_params(capi, k)
    want = np.asarray(k["tx_delay"], dtype=np.float32).ravel()
    assert np.array_equal(ob.us_tx_delays(p), want)
    # host-only helper of the C-ABI (no device needed)
    lib = capi.load_library()
    tx = np.empty(len(want), np.float32)
    assert lib.pbrt_us_tx_delays(C.byref(p), tx.ctypes.data) == 0
    assert np.array_equal(tx, want)
    assert want.reshape(5, 64)[0, 0] == pytest.approx(6.610378e-07, rel=1e-6)
    assert want.reshape(5, 64)[0, 63] == pytest.approx(-6.610378e-07, rel=1e-6)
    assert np.all(want.reshape(5, 64)[2] == 0)
    assert k["elem_x"][0] == pytest.approx(-0.00378, rel=1e-6) and k["elem_x"][63] == pytest.approx(0.00378, rel=1e-6)


def test_k2_plugin_elem_x(mi, known):
    k = known["K2_tx_delay"]
    ui = mi.UltraIntegrator(mi.Properties("ultrasound_integrator", dict(n_elements=64, pitch=1.2e-4)))
    assert np.allclose(ui.elem_x, k["elem_x"], rtol=0, atol=1e-9)


def test_k3_attenuation(ob, known):
    k = known["K3_attenuation"]
    for d, f in zip(k["distance"], k["factor"]):
        assert ob.us_attenuation(k["attenuation"], k["frequency"], d) == pytest.approx(f, abs=2e-7)


def test_k4_directivity_trapezoid(ob, known):
    k = known["K4_directivity"]
    for a, w in zip(k["angle_deg"], k["weight"]):
        assert ob.us_directivity_i(a, k["main_beam_angle"], k["cutoff_angle"]) == pytest.approx(w, abs=2e-5)


def test_k5_impedance_coefficients(ob, known):
    k = known["K5_impedance"]
    for r in k["rows"]:
        got = ob.us_impedance(k["Z1"], k["Z2"], r["cosTr"])
        assert bool(got[4]) == r["tir"]
        if not r["tir"]:
            assert got[0] == pytest.approx(r["Ar"], abs=2e-5)   # Ar is ill-conditioned near TIR: f32 vs f64
            assert got[1] == pytest.approx(r["At"], abs=2e-5)
            assert got[2] == pytest.approx(r["Ar2"], abs=4e-5)
        assert got[3] == pytest.approx(r["pdf_reflect"], rel=1e-6)
    thr = k["tir_cos_threshold"]
    assert thr == pytest.approx(0.98809481, abs=1e-8)
    assert bool(ob.us_impedance(k["Z1"], k["Z2"], thr - 1e-4)[4]) and not bool(ob.us_impedance(k["Z1"], k["Z2"], thr + 1e-4)[4])


def test_k5_through_ultra_bsdf_sample(ob, capi):
    """Normal incidence on a mirror-smooth facet: sample1 = 0.5 puts the micro-normal on the macro normal, so
    UltraBSDF.sample must return Ar = 0.7333 (reflect) / At = 0.2667 (transmit) and pdf_reflect = 0.25."""
    m = capi.make_material(capi.MAT_ULTRA, [7.8, 0.5, 1.2])
    wi = np.array([[0, 0, 1.0], [0, 0, 1.0]], np.float32)
    wo, pdf, w, lobe = ob.bsdf_sample(m, capi.USQ_REFERENCE, wi, [0, 0, 1], [0, 0, 1], 0.5, np.array([[0.1, 0], [0.9, 0]]))
    assert list(lobe) == [0, 1]
    assert w[0, 0] == pytest.approx(0.7333333, abs=1e-6) and w[1, 0] == pytest.approx(0.2666667, abs=1e-6)
    assert pdf[0] == pytest.approx(0.25, abs=1e-6)


def test_k6_put_data(ob, capi, known):
    k = known["K6_put_data"]
    r = capi.UsReceiver()
    r.number_of_elements, r.pitch, r.sample_rate, r.time_samples = (k["number_of_elements"], k["pitch"], k["sample_rate"],
                                                                    k["time_samples"])
    rays = k["rays"]
    buf = np.zeros((r.number_of_elements, r.time_samples), np.float32)
    ob.us_put_data(r, [x["x"] for x in rays], [x["time"] for x in rays], np.array([x["d"] for x in rays], np.float32),
                   [x["amplitude"] for x in rays], buf)
    nz = {(int(i), int(j)): float(buf[i, j]) for i, j in np.argwhere(buf != 0)}
    want = {(e["element"], e["sample"]): e["value"] for e in k["nonzero"]}
    assert nz.keys() == want.keys()
    for key in want:
        assert nz[key] == pytest.approx(want[key], abs=1e-6)
    assert (0, 10) in want and (2, 15) in want and (4, 5) in want and len(want) == 3   # the 4th ray (x = 10) is dropped
    assert want[(4, 5)] == pytest.approx(0.78086881, abs=1e-7)
