"""CPU test against the reference's own run on HALF- and WHOLE-pixel lags under unrotated headers
(tests/golden/make_golden_half_pixel.py -> half_pixel_golden.{npz,json}: four scenes with 3 % NaN pixels x both branches x
orders 1, 2, 3): the lags that bring whole rows and columns of coordinates back within wcslib's rounding noise of k + 1/2
(an even order changes its footprint there) and of k (odd orders; the bounds rule at every order).  The oracle -- whose
coordinates near integers and bounds are wcslib's bit for bit -- reproduces the maps to the float32-rounding level.
GPU: tests/test_gpu_reference_half_pixel.py."""
import numpy as np
import pytest

from tests import golden_cases as G

F = "half_pixel_golden"


@pytest.mark.parametrize("name", G.case_names("corr", F))
def test_oracle_reproduces_the_reference_map(name):
    want, c = G.expected(name, F)
    got = G.oracle_replay(name, counts=2 if c["ctor"]["parallelism"] else None, fixture=F)
    assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1.1e-9, np.nanmax(np.abs(got - want))
    assert np.nanargmax(got) == np.nanargmax(want)
