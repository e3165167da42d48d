"""GPU test against the reference's own run on half- and whole-pixel lags under unrotated headers
(tests/golden/half_pixel_golden.{npz,json}; generator tests/golden/make_golden_half_pixel.py): every case through the
drop-in `Alignment` on FITS files, 1e-7, identical NaN pattern and argmax."""
import warnings

import numpy as np
import pytest

from tests import golden_cases as G

pytestmark = pytest.mark.gpu

F = "half_pixel_golden"


@pytest.fixture(scope="module")
def fits_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("reference_half_pixel")


@pytest.mark.parametrize("name", G.case_names("corr", F))
def test_hip_path_reproduces_the_reference_map(name, fits_dir):
    want, c = G.expected(name, F)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, got = G.product_replay(name, fits_dir, fixture=F)
    assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-7, np.nanmax(np.abs(got - want))
    assert np.nanargmax(got) == np.nanargmax(want)
