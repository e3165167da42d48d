"""GPU tests of the plate-carree path (Alignment.align_using_initial_carrington, reference alignment.py:344-399): inputs
that are already Carrington maps, per-lag sphere rotation between the two maps (oblique CAR for CRVAL2 lags)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _sweep(h, small, hs, large, hl, lags, order=2, method=0, **kw):
    from euispice_coreg_amd import _lib
    h.set_small(small)
    h.set_reference_on_grid(np.asarray(large, dtype=np.float32))
    ls = _lib.LagSet(*lags)
    return h.sweep_helioprojective(hl, hs, ls, order=order, method=method, **kw).reshape(ls.shape + (1,))


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("use_lds", [1, 0])
def test_sweep_car_vs_oracle(gpu_handle, order, use_lds):
    from euispice_coreg_amd import synthetic
    small, hs, large, hl, truth = synthetic.make_car_scene()
    lags = (np.arange(-0.01, 0.045, 0.006), np.arange(-0.035, 0.02, 0.006), None, None, [0.0, 0.4])
    gpu_handle.set_option("use_lds", use_lds)
    try:
        got = _sweep(gpu_handle, small, hs, large, hl, lags, order=order)
    finally:
        gpu_handle.set_option("use_lds", 1)
    want = H.oracle_helio(small.astype(np.float64), hs, large.astype(np.float64), hl, lags, order=order,
                          parallelism=False, unit_lag="deg")
    H.assert_corr_close(got, want, 1e-7, f"CAR sweep order={order} lds={use_lds}")
    am = np.unravel_index(np.nanargmax(got), got.shape)
    assert abs(lags[0][am[0]] - truth["lag_crval1"]) <= 0.006 and abs(lags[1][am[1]] - truth["lag_crval2"]) <= 0.006


def test_sweep_car_cdelt_crota_slices_and_residus(gpu_handle):
    from euispice_coreg_amd import synthetic
    small, hs, large, hl, _ = synthetic.make_car_scene(small_shape=(70, 96), large_shape=(110, 150), crota=0.3, seed=5,
                                                       small_cdelt=(0.0111, 0.0093), nan_frac=0.0)
    lags = (np.array([0.0, 0.012, 0.02]), np.array([-0.015, -0.004]), [0.0, 0.0002], [0.0, -0.0001], [-0.2, 0.0])
    got = _sweep(gpu_handle, small, hs, large, hl, lags)
    want = H.oracle_helio(small.astype(np.float64), hs, large.astype(np.float64), hl, lags, parallelism=False,
                          unit_lag="deg")
    H.assert_corr_close(got, want, 1e-7, "CAR 5-D sweep")
    # contiguous slices of the raveled lag range concatenate to the full map
    from euispice_coreg_amd import _lib
    ls = _lib.LagSet(*lags)
    parts = [gpu_handle.sweep_helioprojective(hl, hs, ls, lag_begin=a, lag_end=b) for a, b in ((0, 17), (17, 18), (18, 48))]
    assert np.allclose(np.concatenate(parts), got.ravel(), rtol=0, atol=1e-12, equal_nan=True)


def test_car_invalid_pole_lags_are_nan(gpu_handle):
    """A header with an explicit LONPOLE = 0 (what astropy's to_header writes for an equatorial CAR map) has no valid
    native pole once a lag makes CRVAL2 negative: astropy raises, the reference's worker dies; here those lag-points are
    NaN and the others are unaffected."""
    from euispice_coreg_amd import synthetic
    small, hs, large, hl, _ = synthetic.make_car_scene(explicit_lonpole=True, pointing_error=(0.018, 0.011), seed=8)
    lags = (np.array([0.0, 0.018]), np.array([-0.02, -0.0050, 0.011, 0.02]), None, None, None)
    got = _sweep(gpu_handle, small, hs, large, hl, lags)
    want = H.oracle_helio(small.astype(np.float64), hs, large.astype(np.float64), hl, lags, parallelism=False,
                          unit_lag="deg")
    # CRVAL2 of the handed-out header is 0.00037 - 0.011: lags below +0.01063 leave it negative
    assert np.isnan(want[:, :2]).all() and np.isfinite(want[:, 2:]).all()
    H.assert_corr_close(got, want, 1e-7, "CAR invalid-pole lags")


def test_align_using_initial_carrington_dropin(tmp_path):
    from euispice_coreg_amd import synthetic
    from euispice_coreg_amd.hdrshift import Alignment, AlignmentResults
    from euispice_coreg_amd.utils import fits_io
    small, hs, large, hl, truth = synthetic.make_car_scene()
    ps, pl = str(tmp_path / "small_car.fits"), str(tmp_path / "large_car.fits")
    fits_io.write_images(ps, [(None, {}), (small, hs)])
    fits_io.write_images(pl, [(None, {}), (large, hl)])
    lag1, lag2 = np.arange(-25.0, 135.0, 20.0), np.arange(-125.0, 35.0, 20.0)  # arcsec; the maps are in degrees
    A = Alignment(pl, ps, lag_crval1=lag1, lag_crval2=lag2, lag_cdelt1=None, lag_cdelt2=None, lag_crota=None,
                  parallelism=True, small_fov_value_max=2900.0)
    with pytest.warns(UserWarning):
        res = A.align_using_initial_carrington()
    assert isinstance(res, AlignmentResults) and res.corr.shape == (8, 8, 1, 1, 1, 1) and res.unit_lag == "deg"
    sm = small.astype(np.float64)
    from oracle import coreg_oracle as O
    O.set_threshold_minmax_to_nan(sm, None, 2900.0)
    # both branches of the reference build the sub-map for this frame (alignment.py:649-651, 765-767): the oracle's frame
    # "initial_carrington"; pinned by the reference's own output in tests/test_gpu_reference_golden.py
    st = H.oracle_state(sm, hs, large.astype(np.float64), hl, (lag1 / 3600.0, lag2 / 3600.0, None, None, None),
                        unit_lag="deg")
    want = O.find_best_header_parameters(st, "initial_carrington", use_ang2pipi=False)
    H.assert_corr_close(res.corr, want, 1e-7, "Alignment.align_using_initial_carrington")
    assert abs(res.shift_arcsec[0] - truth["lag_crval1"] * 3600) < 20 and abs(res.shift_arcsec[1] - truth["lag_crval2"] * 3600) < 20
    # TAN inputs are refused
    from tests import helpers
    s2, h2, l2, hl2, _ = helpers.scene(small_n=32, large_n=48)
    with pytest.raises(ValueError):
        Alignment((l2, hl2), (s2, h2), [0.0], [0.0], None, None, None).align_using_initial_carrington()


@pytest.mark.parametrize("seed", list(range(300, 312)))
def test_fuzz_car(gpu_handle, seed):
    """Random Carrington-map pairs: rolls, unequal / negative pixel sizes, reference latitudes on both sides of the
    equator and far from it, lag sets that cross CRVAL2 = 0, both spline orders, both methods' NaN rules."""
    from euispice_coreg_amd import synthetic
    rng = np.random.default_rng(seed)
    ny, nx = int(rng.integers(40, 100)), int(rng.integers(40, 100))
    cd = (0.0101 * rng.uniform(0.8, 1.3) * rng.choice([1.0, -1.0]), 0.0099 * rng.uniform(0.8, 1.3))
    small, hs, large, hl, _ = synthetic.make_car_scene(small_shape=(ny, nx), large_shape=(int(ny * 1.4), int(nx * 1.5)),
                                                       seed=seed, small_cdelt=(abs(cd[0]), cd[1]),
                                                       crota=float(rng.choice([0.0, 0.4, -2.5, 30.0])),
                                                       nan_frac=float(rng.choice([0.0, 0.01])),
                                                       explicit_lonpole=bool(rng.integers(0, 2)))
    hs = dict(hs)
    if cd[0] < 0:  # flip the longitude axis of the map to align (header only: the oracle and the GPU see the same data)
        hs["CDELT1"] = cd[0]
        lam = hs["CDELT2"] / hs["CDELT1"]
        rho = np.deg2rad(hs["CROTA"])
        hs["PC1_2"], hs["PC2_1"] = float(-lam * np.sin(rho)), float(np.sin(rho) / lam)
    off = float(rng.choice([0.0, 0.0, 12.0, -33.0]))  # move both maps away from the equator
    if "LONPOLE" in hl:
        off = abs(off)  # an explicit LONPOLE = 0 is only valid north of the equator (the library refuses such a target)
    hs["CRVAL2"] += off
    hl = dict(hl, CRVAL2=hl["CRVAL2"] + off)
    l1 = np.sort(rng.uniform(-0.05, 0.05, int(rng.integers(1, 6))))
    l2 = rng.uniform(-0.04, 0.04, int(rng.integers(1, 6)))
    crot = [0.0] if rng.integers(0, 2) else [-0.3, 0.0, 0.2]
    cdl = None if rng.integers(0, 2) else [0.0, 0.0002]
    lags = (l1, l2, cdl, None, crot)
    order = int(rng.choice([1, 2]))
    got = _sweep(gpu_handle, small, hs, large, hl, lags, order=order)
    want = H.oracle_helio(small.astype(np.float64), hs, large.astype(np.float64), hl, lags, order=order,
                          parallelism=False, unit_lag="deg")
    H.assert_corr_close(got, want, 1e-7, f"fuzz CAR seed={seed}")


def _sub_map_scene(seed, force_order=None):
    """One random pair of Carrington maps, lag axes through exactly zero: (small, hs, large, hl, lags, order)."""
    from euispice_coreg_amd import synthetic
    rng = np.random.default_rng(seed)
    ny, nx = int(rng.integers(40, 90)), int(rng.integers(40, 90))
    cd = (0.0101 * rng.uniform(0.8, 1.3) * rng.choice([1.0, -1.0]), 0.0099 * rng.uniform(0.8, 1.3))
    small, hs, large, hl, _ = synthetic.make_car_scene(small_shape=(ny, nx), large_shape=(int(ny * 1.5), int(nx * 1.6)),
                                                       seed=seed, small_cdelt=(abs(cd[0]), cd[1]),
                                                       crota=float(rng.choice([0.0, 0.4, -2.5])),
                                                       nan_frac=float(rng.choice([0.0, 0.01])),
                                                       explicit_lonpole=bool(rng.integers(0, 2)))
    hs = dict(hs)
    if cd[0] < 0:
        hs["CDELT1"] = cd[0]
        lam, rho = hs["CDELT2"] / hs["CDELT1"], np.deg2rad(hs["CROTA"])
        hs["PC1_2"], hs["PC2_1"] = float(-lam * np.sin(rho)), float(np.sin(rho) / lam)
    off = float(rng.choice([0.0, 9.0, -21.0]))
    if "LONPOLE" in hl:
        off = abs(off) + 0.5  # an explicit LONPOLE = 0 is only valid north of the equator
    hs["CRVAL2"] += off
    hl = dict(hl, CRVAL2=hl["CRVAL2"] + off)
    l1 = np.unique(np.concatenate([[0.0], np.round(rng.uniform(-0.03, 0.03, int(rng.integers(1, 4))), 4)]))
    l2 = np.unique(np.concatenate([[0.0], np.round(rng.uniform(0.0, 0.03, int(rng.integers(1, 3))), 4)]))
    lags = (l1, l2, None, None, [0.0] if rng.integers(0, 2) else [0.0, 0.3])
    order = int(rng.choice([1, 2]))
    if force_order is not None:
        order = int(force_order)
    return small, hs, large, hl, lags, order


def _sub_map_case(gpu_handle, seed, force_order=None):
    """The pair through the sub-map semantics: (GPU map, oracle map, lags, order, header of the map to align)."""
    from euispice_coreg_amd import _lib
    from oracle import coreg_oracle as O
    small, hs, large, hl, lags, order = _sub_map_scene(seed, force_order)
    ls = _lib.LagSet(*lags)
    gpu_handle.set_small(small)
    gpu_handle.prepare_reference_helioprojective(large, hl, hs, order)
    got = gpu_handle.sweep_helioprojective(hs, hs, ls, order=order).reshape(ls.shape + (1,))
    st = H.oracle_state(small.astype(np.float64), hs, large.astype(np.float64), hl, lags, order=order, unit_lag="deg")
    want = O.find_best_header_parameters(st, "initial_carrington", use_ang2pipi=False)
    assert np.isfinite(want).any()
    return got, want, lags, order, hs


@pytest.mark.parametrize("seed", list(range(500, 512)))
def test_fuzz_initial_carrington_sub_map_semantics_with_the_identity_lag(gpu_handle, seed):
    """Random pairs of Carrington maps through the semantics `align_using_initial_carrington` really has (round 5: the
    sub-map of alignment.py:987-1016 in both branches -- reference map resampled once on the grid of the map to align by
    `coreg_prepare_reference_helioprojective` with two CAR headers, every lag on that grid), lag axes THROUGH exactly
    zero: the identity lag-point is swept with the exact identity map and wcslib's CAR chain decides its border pixels
    (and, for order 1, the tap set of every pixel).  Rolls, unequal / negative pixel sizes, both hemispheres, explicit
    LONPOLE, NaN pixels, both spline orders; against the oracle's frame "initial_carrington"."""
    got, want, lags, order, _ = _sub_map_case(gpu_handle, seed)
    H.assert_corr_close(got, want, 1e-7, f"CAR sub-map fuzz seed={seed} order={order}")


@pytest.mark.parametrize("seed", [1143, 1325, 1403, 1412, 1934, 2455])
def test_pure_latitude_lags_of_unrotated_carrington_maps_at_order_1(gpu_handle, seed):
    """What tests/deep_fuzz_car.py met at the end of round 5 (6 of 3 000 cases): two UNROTATED Carrington maps, an odd
    spline order, and a lag in CRVAL2 alone.  Columns then map to columns -- x' comes back within wcslib's rounding noise
    of the integer i for every pixel -- and the sign of that noise picks the taps of the order-1 spline, hence which
    neighbour's NaN poisons the sample (3e-6 .. 1.7e-3 on the coefficient of those lag-points).  The plate-carree sweep
    now runs the single-sample pass of DESIGN 4b as the helioprojective sweeps do (scan with the sphere-rotation map over
    the lags that keep an image axis invariant, `WcslibCar` on the host, `k_tap_fix<.., MODE_CAR>`): within the tolerance,
    and the deviation is back when the pass is switched off."""
    got, want, lags, order, hs = _sub_map_case(gpu_handle, seed)
    assert order == 1 and hs["CROTA"] == 0.0
    H.assert_corr_close(got, want, 1e-7, f"CAR sub-map seed={seed}")
    gpu_handle.set_option("tap_fix", 0)
    try:
        raw, _, _, _, _ = _sub_map_case(gpu_handle, seed)
    finally:
        gpu_handle.set_option("tap_fix", 1)
    d = np.abs(raw - want)[..., 0]
    l1, l2, lr = lags[0], lags[1], np.asarray(lags[4])
    kind = (l1[:, None, None, None, None] == 0.0) & (l2[None, :, None, None, None] != 0.0) & \
           (lr[None, None, None, None, :] == 0.0)
    kind = np.broadcast_to(kind, d.shape)
    assert d[~kind].max() <= 1e-7 and 1e-7 < d[kind].max() < 5e-3, d


def test_car_invalid_target_header_is_an_error(gpu_handle):
    from euispice_coreg_amd import _lib, synthetic
    small, hs, large, hl, _ = synthetic.make_car_scene(explicit_lonpole=True, small_shape=(40, 40), large_shape=(60, 60))
    hl = dict(hl, CRVAL2=-20.0)  # LONPOLE = 0 south of the equator: astropy cannot build WCS(hdr_large) either
    with pytest.raises(_lib.CoregError):
        _sweep(gpu_handle, small, hs, large, hl, ([0.0], [0.0], None, None, None))


@pytest.mark.parametrize("dlat,use_lds,tile_w", [(0.6, 1, 0), (0.6, 0, 0), (-0.8, 1, 256), (0.6, 1, 4)])
def test_car_maps_with_different_reference_latitudes(gpu_handle, dlat, use_lds, tile_w):
    """The map to align is referenced at a different latitude than the target map (oblique relation between the two
    plate-carree grids: the per-lag map is a sphere rotation followed by atan2 / asin, not projective, so a tile's image
    is not bounded by its mapped corners -- the sweep widens the corner boxes by a curvature margin).  Parity with the
    oracle, NaN pattern included, for square, 256 x 4 and 4 x 256 tiles and for the global-memory gather."""
    from euispice_coreg_amd import synthetic
    small, hs, large, hl, _ = synthetic.make_car_scene(small_shape=(96, 128), large_shape=(180, 240), seed=11,
                                                       crota=0.4, nan_frac=0.004)
    hs = dict(hs, CRVAL2=hs["CRVAL2"] + dlat)
    lags = (np.arange(-0.02, 0.03, 0.01), np.arange(-0.03, 0.02, 0.01), None, None, [0.0, 0.5])
    gpu_handle.set_option("use_lds", use_lds)
    gpu_handle.set_option("tile_w", tile_w)
    try:
        got = _sweep(gpu_handle, small, hs, large, hl, lags)
    finally:
        gpu_handle.set_option("use_lds", 1)
        gpu_handle.set_option("tile_w", 0)
    want = H.oracle_helio(small.astype(np.float64), hs, large.astype(np.float64), hl, lags, parallelism=False,
                          unit_lag="deg")
    assert np.isfinite(want).any()
    H.assert_corr_close(got, want, 1e-7, f"CAR dlat={dlat} lds={use_lds} tile_w={tile_w}")


@pytest.mark.parametrize("use_lds", [1, 0])
def test_sweep_car_on_a_full_sun_map_up_to_the_pole(gpu_handle, use_lds):
    """Full-Sun plate-carree reference (latitudes -89.75 .. 89.75) and a map to align whose rows run from latitude 30 to
    89.6: towards the pole the longitude of the rotated frame turns quickly, the images of a tile's corners no longer
    bound the tile's image, and the kernel must not trust a staged window there (car_tile_margin: polar tiles take the
    per-point path).  LDS and global-memory gathers against the oracle, CRVAL2 lags make the shifted maps oblique."""
    rng = np.random.default_rng(77)

    def hdr(nx, ny, crval, cdelt, crpix=None, rot=0.0):
        rho, lam = np.deg2rad(rot), cdelt[1] / cdelt[0]
        crpix = crpix or ((nx + 1) / 2.0, (ny + 1) / 2.0)
        return {"NAXIS": 2, "NAXIS1": nx, "NAXIS2": ny, "CTYPE1": "CRLN-CAR", "CTYPE2": "CRLT-CAR", "CUNIT1": "deg",
                "CUNIT2": "deg", "CRPIX1": crpix[0], "CRPIX2": crpix[1], "CRVAL1": crval[0], "CRVAL2": crval[1],
                "CDELT1": cdelt[0], "CDELT2": cdelt[1], "PC1_1": float(np.cos(rho)),
                "PC1_2": float(-lam * np.sin(rho)), "PC2_1": float(np.sin(rho) / lam), "PC2_2": float(np.cos(rho)),
                "CROTA": rot, "DATE-AVG": "2022-03-17T09:50:45.277"}

    hl = hdr(720, 360, (180.0, 0.0), (0.5, 0.5))
    hs = hdr(240, 150, (200.0, 0.3), (0.4, 0.4), crpix=(120.5, -74.0), rot=1.5)  # rows at latitude ~30 .. 89.6
    yy, xx = np.mgrid[0:360, 0:720]
    large = (300.0 + 200.0 * np.sin(xx / 17.0) * np.cos(yy / 11.0) + 20.0 * rng.standard_normal((360, 720))).astype(np.float32)
    ys, xs = np.mgrid[0:150, 0:240]
    small = (300.0 + 200.0 * np.sin(xs / 13.0 + 0.4) * np.cos(ys / 9.0) + 20.0 * rng.standard_normal((150, 240))).astype(np.float32)
    small[rng.random(small.shape) < 0.003] = np.nan
    lags = (np.array([-1.0, 0.0, 0.7, 1.4]), np.array([-0.6, 0.0, 0.9]), None, None, [0.0, 0.5])
    gpu_handle.set_option("use_lds", use_lds)
    try:
        got = _sweep(gpu_handle, small, hs, large, hl, lags)
    finally:
        gpu_handle.set_option("use_lds", 1)
    want = H.oracle_helio(small.astype(np.float64), hs, large.astype(np.float64), hl, lags, parallelism=False,
                          unit_lag="deg")
    assert np.isfinite(want).sum() >= 12
    H.assert_corr_close(got, want, 1e-7, f"CAR sweep up to the pole, lds={use_lds}")


def test_a_pure_crota_lag_returns_the_reference_pixel_to_itself(gpu_handle):
    """Seed 3196 of tests/deep_fuzz_car.py: a CROTA lag with both CRVAL lags zero rotates the map about CRPIX, whose pixel
    (integer here) comes back to itself within wcslib's noise -- ONE sample whose order-1 taps the sign decides.  The
    single-sample pass lists it (the scan covers every lag that leaves CRVAL1 or CRVAL2 alone, CROTA / CDELT lags on top
    included): within the tolerance, 4.0e-6 with the pass off."""
    got, want, lags, order, hs = _sub_map_case(gpu_handle, 3196)
    assert order == 1 and gpu_handle.last_tap_fix()["samples"] >= 0
    H.assert_corr_close(got, want, 1e-7, "CAR sub-map seed=3196")
    gpu_handle.set_option("tap_fix", 0)
    try:
        raw, _, _, _, _ = _sub_map_case(gpu_handle, 3196)
    finally:
        gpu_handle.set_option("tap_fix", 1)
    assert 1e-7 < np.nanmax(np.abs(raw - want)) < 1e-4


@pytest.mark.parametrize("order", [1, 3])
def test_grid_shared_plate_carree_launches_keep_their_single_sample_lists(gpu_handle, order):
    """Round 6 (closes DESIGN 9 open 3 of round 5).  A plate-carree sweep is one launch per (CDELT,
    CROTA) combination and every launch lists its noise-decided single samples anew in the handle's buffers; when the
    GRID is shared between GPUs the ill-conditioned lag-points can only be flagged from the REDUCED sums, i.e. after the
    last launch -- and the lists of the earlier launches were gone by then, so such launches were not re-evaluated
    (N > 1 could differ from N = 1).  Each launch's lists are now copied and `coreg_finalize_sums` runs the fix kernels a
    second time about the flagged slots' own pivots, as it does for the one-launch helioprojective sweeps.

    Two UNROTATED Carrington maps, an odd spline order, lags in CRVAL2 alone (columns map to columns: wcslib's noise
    picks the taps) x two CROTA lags = two launches, over data the one-pass moments cannot carry (a nearly flat reference
    with a far-away block that moves the pivot): every lag-point is flagged.  Two and three emulated ranks give the
    one-GPU map to 1e-10 and the two-pass oracle to 1e-7; with the single-sample pass off they do not."""
    from euispice_coreg_amd import _lib
    from oracle import coreg_oracle as O
    rng = np.random.default_rng(17)
    # the pair of tests/deep_fuzz_car.py's seed 1143 (unrotated, 1 % NaN pixels; 3e-6 .. 1.7e-3 on its pure-latitude
    # lag-points without the single-sample pass), its reference swapped for the ill-conditioned one
    small, hs, _, _, lags0, _ = _sub_map_scene(1143)
    assert hs["CROTA"] == 0.0 and np.isnan(small).any()
    small = np.array(small, dtype=np.float64)
    small[:8, :8] = np.nan
    ref = (1000.0 + 1.0e-3 * rng.standard_normal(small.shape)).astype(np.float64)
    ref[:6, :6] = -1.0e7
    l2 = np.unique(np.concatenate([np.asarray(lags0[1]), [0.0113]]))
    lags = ([0.0], l2, None, None, [0.0, 0.3])
    ls = _lib.LagSet(*lags)
    st = H.oracle_state(small, hs, ref, hs, lags, order=order, unit_lag="deg")
    want = O.find_best_header_parameters(st, "initial_carrington", prepared_reference=ref, use_ang2pipi=False)[..., 0]
    want = want.reshape(ls.shape)
    assert np.isfinite(want).all() and np.abs(want).max() < 0.3
    gpu_handle.set_small(small)
    gpu_handle.set_reference_on_grid(ref)
    got = gpu_handle.sweep_helioprojective(hs, hs, ls, order=order).reshape(ls.shape)
    counts = gpu_handle.last_visit_counts()
    assert counts["refined_lag_points"] == ls.size and counts["flagged_not_refined"] == 0, counts
    assert np.abs(got - want).max() <= 1e-7, np.abs(got - want)

    def shared(world):
        total = None
        try:
            for r in range(world):
                gpu_handle.set_point_shard(r, world)
                gpu_handle.sweep_helioprojective(hs, hs, ls, order=order)
                part = gpu_handle.copy_sums()
                total = part if total is None else total + part
            return gpu_handle.finalize_sums(total, ls.size).reshape(ls.shape)
        finally:
            gpu_handle.set_point_shard(0, 1)

    for world in (2, 3):
        m = shared(world)
        assert gpu_handle.last_visit_counts()["refined_lag_points"] == ls.size
        assert np.abs(m - want).max() <= 1e-7 and np.abs(m - got).max() <= 1e-10, (world, np.abs(m - got))
    # the single samples matter at this level: without the pass the pure-latitude lag-points of the first launch are off
    gpu_handle.set_option("tap_fix", 0)
    try:
        nofix = shared(2)
    finally:
        gpu_handle.set_option("tap_fix", 1)
    d = np.abs(nofix - want)
    assert d[0, 1:, 0, 0, 0].max() > 1e-7, d   # (l2[0] = 0 is the identity lag: k_parity_fix's)
