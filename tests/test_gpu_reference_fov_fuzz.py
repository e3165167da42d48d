"""GPU tests against the RANDOM family of reference-run fixtures for `fov_limits` / `remove_fov_limits`
(tests/golden/fov_fuzz_golden.{npz,json}; generator: tests/golden/make_golden_fov_fuzz.py, which ran the reference's own
`Alignment.align_using_helioprojective` with random boxes on 8 seeded random scenes x 2 calls).  Every case goes through
the drop-in `euispice_coreg_amd.hdrshift.Alignment` on FITS files -- the box masks and the sub-FOV re-grid of the image to
align (alignment.py:844-874, 1082-1127) on the host side of the product, the sweep on the re-gridded image through the C
ABI -- within 1e-7, identical NaN pattern and argmax."""
import warnings

import numpy as np
import pytest

from tests import golden_cases as G

pytestmark = pytest.mark.gpu

F = "fov_fuzz_golden"


@pytest.fixture(scope="module")
def fits_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("reference_fov_fuzz")


@pytest.mark.parametrize("name", G.case_names("corr", F))
def test_hip_path_reproduces_the_reference_map(name, fits_dir):
    want, c = G.expected(name, F)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, got = G.product_replay(name, fits_dir, fixture=F)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN pattern"
    d = np.abs(got - want)
    print(f"{name}: max |HIP - reference| = {np.nanmax(d):.2e}")
    assert np.nanmax(d) <= 1e-7, f"max |HIP - reference| = {np.nanmax(d):.3e}"
    am = np.nanargmax(got)
    assert am == np.nanargmax(want) or want.ravel()[am] >= np.nanmax(want) - 1e-7
