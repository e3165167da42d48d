"""CPU tests against the RANDOM family of reference-run fixtures for the PLATE-CARREE frame
(tests/golden/make_golden_car_fuzz.py ran euispice_coreg.hdrshift.Alignment.align_using_initial_carrington on 10 seeded
random pairs of Carrington maps x 2 calls: rolled / unrolled maps, unequal and negative pixel sizes, both hemispheres and
the equator, an explicit LONPOLE, NaN fractions, CRVAL lags in degrees through and around zero, CROTA lags, orders 1-3,
both branches, a threshold).  The oracle restates the reference's chain through its CAR restatement of wcslib
(`oracle.CarWCS` + `WcslibCar` for the noise-decided samples); samples are rounded to float32 in this frame
(alignment.py:1024), so the bound is test_reference_golden_cpu.py's 1.1e-9.  GPU: tests/test_gpu_reference_car_fuzz.py."""
import numpy as np
import pytest

from tests import golden_cases as G

F = "car_fuzz_golden"


def test_fixture_is_what_the_generator_describes():
    g, m = G.load(F)
    assert m["interpreter"]["astropy"] == "4.3.1" and m["interpreter"]["seed"] == 91000
    assert len(m["scenes"]) == 10 and len(m["cases"]) == 20 and len(G.case_names("corr", F)) == 20
    calls = [(c["ctor"]["parallelism"], c["ctor"]["reprojection_order"]) for c in m["cases"].values()]
    assert {o for _, o in calls} == {1, 2, 3} and {p for p, _ in calls} == {True, False}
    assert all(c["call"] == "initial_carrington" and c["ctor"]["unit_lag"] == "deg" for c in m["cases"].values())
    hs = [s["hdr_small"] for s in m["scenes"].values()]
    assert all(h["CTYPE1"] == "CRLN-CAR" for h in hs)
    assert any(h["CDELT1"] < 0 for h in hs) and any(h["CROTA"] == 0.0 for h in hs) and any(h["CROTA"] != 0.0 for h in hs)
    assert any(h["CRVAL2"] > 5 for h in hs) and any(h["CRVAL2"] < -5 for h in hs) and any(abs(h["CRVAL2"]) < 1 for h in hs)
    assert any("LONPOLE" in s["hdr_large"] for s in m["scenes"].values())
    through_zero = sum(0.0 in c["ctor"]["lag_crval1"] and 0.0 in c["ctor"]["lag_crval2"] for c in m["cases"].values())
    assert through_zero >= 3  # identity lags: border pixels / tap sets decided by wcslib's CAR chain
    assert sum(c["ctor"]["lag_crota"] is not None for c in m["cases"].values()) >= 6


@pytest.mark.parametrize("name", G.case_names("corr", F))
def test_oracle_reproduces_the_reference_map(name):
    want, c = G.expected(name, F)
    got = G.oracle_replay(name, counts=2 if c["ctor"]["parallelism"] else None, fixture=F)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN pattern"
    d = np.abs(got - want)
    assert np.nanmax(d) <= 1.1e-9, f"max |oracle - reference| = {np.nanmax(d):.3e}"
    assert np.nanargmax(got) == np.nanargmax(want)
