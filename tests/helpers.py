"""Shared helpers of the parity tests: small seeded scenes, oracle sweeps, GPU sweeps through the C ABI."""
import numpy as np

from euispice_coreg_amd import synthetic
from oracle import coreg_oracle as O

CARR_LON = (228.0, 262.0)
CARR_LAT = (-12.0, 22.0)


def scene(small_n=96, large_n=160, seed=5, **kw):
    return synthetic.make_scene(small_n=small_n, large_n=large_n, seed=seed, n_blobs=kw.pop("n_blobs", 120), **kw)


def oracle_state(small, hdr_small, large, hdr_large, lags, order=2, unit_lag="arcsec", shape=None, lonlims=None,
                 latlims=None, solar_r=None, cdelt_semantics="intended"):
    hs, hl = dict(hdr_small), dict(hdr_large)
    O.check_and_create_pcij_matrix(hs)
    O.check_and_create_pcij_matrix(hl)
    st = O.SweepState(hs, hl, small, large, lags[0], lags[1], lags[2], lags[3], lags[4], lag_solar_r=solar_r,
                      unit_lag=unit_lag, order=order, cdelt_semantics=cdelt_semantics)
    st.shape, st.lonlims, st.latlims = shape, lonlims, latlims
    return st


def oracle_carrington(small, hdr_small, large, hdr_large, lags, shape, lonlims=CARR_LON, latlims=CARR_LAT, order=2,
                      solar_r=(1.004,), counts=None, cdelt_semantics="intended"):
    st = oracle_state(small, hdr_small, large, hdr_large, lags, order=order, shape=list(shape), lonlims=list(lonlims),
                      latlims=list(latlims), solar_r=solar_r, cdelt_semantics=cdelt_semantics)
    return O.find_best_header_parameters(st, "carrington", counts=counts)


def oracle_helio(small, hdr_small, large, hdr_large, lags, order=2, parallelism=True, unit_lag="arcsec", counts=None,
                 cdelt_semantics="intended"):
    st = oracle_state(small, hdr_small, large, hdr_large, lags, order=order, unit_lag=unit_lag,
                      cdelt_semantics=cdelt_semantics)
    return O.find_best_header_parameters(st, "helioprojective", parallelism=parallelism, counts=counts)


def gpu_carrington(h, small, hdr_small, large, hdr_large, lags, shape, lonlims=CARR_LON, latlims=CARR_LAT, order=2,
                   solar_r=1.004, numpy_lat_trig=True, lag_begin=0, lag_end=None, cdelt_semantics=0,
                   prepare=True):
    from euispice_coreg_amd import _lib
    grid = _lib.Grid(lonlims, latlims, shape, numpy_lat_trig=numpy_lat_trig)
    ls = _lib.LagSet(*lags)
    if prepare:
        h.set_small(small)
        h.prepare_reference_carrington(large, hdr_large, grid, solar_r, order)
    out = h.sweep_carrington(hdr_small, grid, solar_r, ls, order=order, lag_begin=lag_begin, lag_end=lag_end,
                             cdelt_semantics=cdelt_semantics)
    if lag_begin == 0 and lag_end is None:
        return out.reshape(ls.shape + (1,))
    return out


def gpu_helio(h, small, hdr_small, large, hdr_large, lags, order=2, lag_begin=0, lag_end=None, cdelt_semantics=0,
              prepare=True, serial_semantics=False):
    """parallelism=True semantics by default (reference on the small header's own grid, float32);
    serial_semantics: target = full large grid, float64 reference (quirk Q1)."""
    from euispice_coreg_amd import _lib
    ls = _lib.LagSet(*lags)
    if prepare:
        h.set_small(small)
        if serial_semantics:
            h.set_reference_on_grid(np.asarray(large, dtype=np.float64))
        else:
            h.prepare_reference_helioprojective(large, hdr_large, hdr_small, order)
    target = hdr_large if serial_semantics else hdr_small
    out = h.sweep_helioprojective(target, hdr_small, ls, order=order, lag_begin=lag_begin, lag_end=lag_end,
                                  cdelt_semantics=cdelt_semantics)
    if lag_begin == 0 and lag_end is None:
        return out.reshape(ls.shape + (1,))
    return out


def assert_corr_close(got, want, atol, what=""):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN pattern differs"
    if np.isfinite(want).any():
        d = np.nanmax(np.abs(got - want))
        assert d <= atol, f"{what}: max|dcorr| = {d:.3e} > {atol:.1e}"
        # same argmax -- or, when several lag-points tie to within the tolerance (e.g. CDELT1 lags under the reference's
        # semantics are no-ops, quirk Q2), an argmax inside the tie
        if np.nanargmax(got) != np.nanargmax(want):
            assert want.ravel()[np.nanargmax(got)] >= np.nanmax(want) - atol, f"{what}: argmax differs"
