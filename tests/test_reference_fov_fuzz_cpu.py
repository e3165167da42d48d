"""CPU tests against the RANDOM family of reference-run fixtures for `fov_limits` / `remove_fov_limits`
(tests/golden/make_golden_fov_fuzz.py ran euispice_coreg.hdrshift.Alignment.align_using_helioprojective with random
sub-FOV boxes and removed boxes on 8 seeded random scenes x 2 calls, arcsec and degree headers, orders 1-3, both
branches).  `fov_limits` re-grids the image to align in FLOAT64 at coordinates from the TAN restatement
(alignment.py:1082-1127) and replaces its header: bound 5e-9 as for the hand-made cases of test_reference_golden_cpu.py;
`remove_fov_limits` alone: 1.1e-9 (float32-rounded samples).  GPU: tests/test_gpu_reference_fov_fuzz.py."""
import numpy as np
import pytest

from tests import golden_cases as G

F = "fov_fuzz_golden"


def test_fixture_is_what_the_generator_describes():
    g, m = G.load(F)
    assert m["interpreter"]["astropy"] == "4.3.1" and m["interpreter"]["seed"] == 93000
    assert len(m["scenes"]) == 8 and len(m["cases"]) == 16 and len(G.case_names("corr", F)) == 16
    kinds = [("fov_limits" in c["call_kwargs"], "remove_fov_limits" in c["call_kwargs"]) for c in m["cases"].values()]
    assert {(True, False), (False, True), (True, True)} == set(kinds)
    assert {c["ctor"]["parallelism"] for c in m["cases"].values()} == {True, False}
    assert {c["ctor"]["reprojection_order"] for c in m["cases"].values()} == {1, 2, 3}
    assert {s["hdr_small"]["CUNIT1"] for s in m["scenes"].values()} == {"arcsec", "deg"}


@pytest.mark.parametrize("name", G.case_names("corr", F))
def test_oracle_reproduces_the_reference_map(name):
    want, c = G.expected(name, F)
    got = G.oracle_replay(name, counts=2 if c["ctor"]["parallelism"] else None, fixture=F)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN pattern"
    tol = 5e-9 if "fov_limits" in c["call_kwargs"] else 1.1e-9
    d = np.abs(got - want)
    assert np.nanmax(d) <= tol, f"max |oracle - reference| = {np.nanmax(d):.3e} > {tol:.1e}"
    assert np.nanargmax(got) == np.nanargmax(want)
