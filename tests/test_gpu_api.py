"""GPU tests of the drop-in API: euispice_coreg_amd.hdrshift.Alignment used exactly like the reference's
(README.md:47-87, 97-139) on synthetic FITS files, checked against the oracle's sweep."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fits_pair(tmp_path_factory):
    from euispice_coreg_amd.utils import fits_io
    d = tmp_path_factory.mktemp("fits")
    small, hs, large, hl, truth = H.scene(small_n=128, large_n=192, seed=11, n_blobs=200)
    ps, pl = str(d / "small.fits"), str(d / "large.fits")
    fits_io.write_images(ps, [(None, {}), (small.astype(np.float32), hs)])
    fits_io.write_images(pl, [(None, {}), (large.astype(np.float32), hl)])
    return ps, pl, small, hs, large, hl, truth


def test_align_using_carrington_dropin(fits_pair):
    from euispice_coreg_amd.hdrshift import Alignment, AlignmentResults
    ps, pl, small, hs, large, hl, truth = fits_pair
    lag1, lag2 = np.arange(5, 30, 4.0), np.arange(-21, 4, 4.0)
    kw = dict(lonlims=(228, 262), latlims=(-12, 22), shape=(96, 96))
    A = Alignment(large_fov_known_pointing=pl, small_fov_to_correct=ps, lag_crval1=lag1, lag_crval2=lag2,
                  lag_cdelt1=[0], lag_cdelt2=[0], lag_crota=[0.0, 0.3], parallelism=True, counts_cpu_max=20,
                  small_fov_value_max=2500.0)
    res = A.align_using_carrington(method="correlation", **kw)
    assert isinstance(res, AlignmentResults) and res.corr.shape == (7, 7, 1, 1, 2, 1)
    sm = small.astype(np.float32).astype(np.float64)
    from oracle import coreg_oracle as O
    O.set_threshold_minmax_to_nan(sm, None, 2500.0)
    want = H.oracle_carrington(sm, hs, large.astype(np.float32).astype(np.float64), hl,
                               (lag1, lag2, [0], [0], [0.0, 0.3]), kw["shape"], kw["lonlims"], kw["latlims"])
    H.assert_corr_close(res.corr, want, 1e-10, "Alignment.align_using_carrington")
    assert tuple(res.max_index) == tuple(np.unravel_index(np.nanargmax(want), want.shape))
    # raw array form
    A2 = Alignment(pl, ps, lag1, lag2, [0], [0], [0.0, 0.3], parallelism=True, small_fov_value_max=2500.0)
    corr = A2.align_using_carrington(return_type="corr", **kw)
    assert np.array_equal(corr, res.corr, equal_nan=True)


@pytest.mark.parametrize("parallelism", [True, False])
def test_align_using_helioprojective_dropin(fits_pair, parallelism):
    from euispice_coreg_amd.hdrshift import Alignment
    ps, pl, small, hs, large, hl, truth = fits_pair
    lag1, lag2 = np.arange(9, 26, 4.0), np.arange(-17, 0, 4.0)
    A = Alignment(pl, ps, lag_crval1=lag1, lag_crval2=lag2, lag_cdelt1=None, lag_cdelt2=None, lag_crota=[0.3],
                  parallelism=parallelism, small_fov_value_min=20.0)
    res = A.align_using_helioprojective(method="correlation")
    sm = small.astype(np.float32).astype(np.float64)
    from oracle import coreg_oracle as O
    O.set_threshold_minmax_to_nan(sm, 20.0, None)
    want = H.oracle_helio(sm, hs, large.astype(np.float32).astype(np.float64), hl, (lag1, lag2, None, None, [0.3]),
                          parallelism=parallelism)
    H.assert_corr_close(res.corr, want, 1e-7, f"Alignment.align_using_helioprojective parallelism={parallelism}")
    assert len(res.shift_arcsec) == 5 and res.unit_lag == "arcsec"


def test_alignment_refuses_unimplemented_paths(fits_pair):
    from euispice_coreg_amd.hdrshift import Alignment
    ps, pl = fits_pair[0], fits_pair[1]
    A = Alignment(pl, ps, [0.0], [0.0], None, None, None)
    with pytest.raises(NotImplementedError):
        A.align_using_helioprojective(method="residus")
    with pytest.raises(NotImplementedError):
        A.align_using_carrington(lonlims=(228, 262), latlims=(-12, 22), shape=(32, 32),
                                 method_carrington_reprojection="sunpy")
