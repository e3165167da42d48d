"""
CPU tests: the oracle (oracle/coreg_oracle.py) against
  * golden vectors produced by the reference's own utils/rectify.py (tests/golden/make_golden_rectify.py),
  * golden vectors produced by astropy.wcs / wcslib (tests/golden/make_golden_wcs.py),
  * the correlation-map fixture embedded in the reference's tests
    (euispice_coreg/hdrshift/test/test_AlignmentResults.py:33-126, known answer :172-173),
  * scipy.ndimage.map_coordinates itself (scalar model of the order-1/2 semantics).
"""
import numpy as np
import pytest

from oracle import coreg_oracle as O
from tests.conftest import golden_header, rectify_case


@pytest.mark.parametrize("case", ["A", "B", "C", "D", "E"])
def test_carrington_coords_bit_equal_reference(rectify_golden, case):
    """CarringtonTransform + Rectifier grid: (nx, ny) bit-identical to the reference (NEP-50 dtype flow, quirk Q6)."""
    c = rectify_case(rectify_golden, case)
    nx, ny = O.carrington_coords(c["hdr"], c["solar_r"], c["shape"], c["lonlims"], c["latlims"])
    assert nx.dtype == np.float64
    assert np.array_equal(nx, c["nx"], equal_nan=True)
    assert np.array_equal(ny, c["ny"], equal_nan=True)


@pytest.mark.parametrize("case", ["A", "B", "C", "D", "E"])
def test_carrington_resample_bit_equal_reference(rectify_golden, case):
    c = rectify_case(rectify_golden, case)
    res = O.carrington_transform_fa(c["image"], c["hdr"], c["solar_r"], c["shape"], c["lonlims"], c["latlims"],
                                    order=c["order"])
    assert np.array_equal(res, c["resampled"], equal_nan=True)


@pytest.mark.parametrize("order", [1, 2])
def test_spline_model_matches_reference_interpol2d_edges(rectify_golden, order):
    """Bounds rule, mirrored edge taps, NaN taps, NaN coordinates (SURVEY a-14, quirk Q11), via the reference's
    rectify.interpol2d run on scipy 1.7.1."""
    g = rectify_golden
    m = O.spline_sample_model(g["edge/image"], g["edge/x"], g["edge/y"], np.nan, order)
    want = g["edge/res_order%d" % order]
    assert np.array_equal(np.isnan(m), np.isnan(want))
    assert np.nanmax(np.abs(m - want)) <= 1e-12
    # float32 destination == float32(float64 result)
    assert np.array_equal(m.astype(np.float32), g["edge/res32_order%d" % order], equal_nan=True)
    # and the oracle's own entry point (scipy of this interpreter)
    got = O.interpol2d(g["edge/image"], g["edge/x"], g["edge/y"], fill=-32762, order=order)
    got = np.where(got == -32762, np.nan, got)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-12


@pytest.mark.parametrize("order", [1, 2])
def test_spline_model_matches_scipy_random(order):
    rng = np.random.default_rng(3)
    img = rng.normal(size=(23, 31))
    img[5, 7] = np.nan
    x = rng.uniform(-2, 33, size=4000)
    y = rng.uniform(-2, 25, size=4000)
    want = O.interpol2d(img, x, y, fill=np.nan, order=order)
    got = O.spline_sample_model(img, x, y, np.nan, order)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-13


@pytest.mark.parametrize("name", ["hri", "fsi", "spice", "far"])
def test_tan_wcs_matches_wcslib(wcs_golden, name):
    g = wcs_golden
    w = O.TanWCS(golden_header(g, name))
    lon, lat = w.pixel_to_world(g[name + "/px"], g[name + "/py"])
    assert np.abs(lon - g[name + "/lon"]).max() <= 1e-12
    assert np.abs(lat - g[name + "/lat"]).max() <= 1e-12
    bx, by = w.world_to_pixel(O.ang2pipi(g[name + "/lon"]), O.ang2pipi(g[name + "/lat"]))
    assert np.abs(bx - g[name + "/back_x"]).max() <= 1e-9
    assert np.abs(by - g[name + "/back_y"]).max() <= 1e-9


@pytest.mark.parametrize("tag", ["lag_hri", "sub_hri_fsi", "lag_spice", "lag_cdelt"])
def test_tan_composite_matches_wcslib(wcs_golden, tag):
    """pixel grid of A -> world -> ang2pipi -> pixels of B (one helioprojective lag-point, alignment.py:1038-1069)."""
    g = wcs_golden
    wa, wb = O.TanWCS(golden_header(g, tag + "/A")), O.TanWCS(golden_header(g, tag + "/B"))
    lon, lat = wa.pixel_to_world(g[tag + "/gx"], g[tag + "/gy"])
    x, y = wb.world_to_pixel(O.ang2pipi(lon), O.ang2pipi(lat))
    assert np.abs(x - g[tag + "/x"]).max() <= 1e-9
    assert np.abs(y - g[tag + "/y"]).max() <= 1e-9


def test_ang2pipi():
    a = np.array([-540.0, -180.0, -179.0, 0.0, 179.0, 180.0, 181.0, 359.0, 360.0, 725.0])
    got = O.ang2pipi(a)
    assert np.all((got > -180.0) & (got <= 180.0))
    assert np.allclose((got - a) % 360.0, 0.0)
    assert got[5] == 180.0 and got[1] == 180.0


def test_c_correlate_is_pearson():
    rng = np.random.default_rng(0)
    a = rng.normal(size=1000) * 30 + 500
    b = 0.3 * a + rng.normal(size=1000) * 10
    r = O.c_correlate(a, b)[0]
    assert abs(r - np.corrcoef(a, b)[0, 1]) < 1e-13
    assert np.isnan(O.c_correlate(np.array([]), np.array([]))[0])


# the 11x6 correlation map of euispice_coreg/hdrshift/test/test_AlignmentResults.py:33-126 (data fixture)
REF_CORR = np.array([
    [0.94431532, 0.94491356, 0.94490277, 0.94429364, 0.94309195, 0.94131598],
    [0.9487374, 0.94936037, 0.94934872, 0.94870775, 0.94744547, 0.94558114],
    [0.95292, 0.95356913, 0.95355487, 0.95288052, 0.95155507, 0.94959962],
    [0.95678181, 0.95745709, 0.95743886, 0.95673169, 0.95534362, 0.95329829],
    [0.96025253, 0.96095169, 0.96093119, 0.96019453, 0.95874962, 0.95662224],
    [0.963255, 0.96397323, 0.96395091, 0.96318901, 0.96169552, 0.95949712],
    [0.96570708, 0.9664386, 0.96641366, 0.96563084, 0.9640988, 0.96184383],
    [0.9675529, 0.96828706, 0.96825363, 0.96745105, 0.96588888, 0.96359088],
    [0.9687609, 0.9694829, 0.96943329, 0.96861061, 0.96702333, 0.96469464],
    [0.96932341, 0.9700199, 0.9699457, 0.96910128, 0.96749419, 0.96514772],
    [0.96927416, 0.96994215, 0.96984541, 0.96898563, 0.96737077, 0.96502305],
]).reshape(11, 6, 1, 1, 1, 1)


def test_compute_shift_reference_fixture():
    """Known answer of the reference's own test (shift_pixels ~ (9.33682107, 1.42187891) +- 1e-2,
    test_AlignmentResults.py:172-173).  scipy.optimize.curve_fit drifts across versions by ~1.2e-2 in x
    (SURVEY section 4): tolerance 2e-2, argmax exact."""
    lag1 = np.arange(15, 26, 1).astype(float)
    lag2 = np.arange(5, 11, 1).astype(float)
    max_index, shift_px, shift_arcsec = O.compute_shift(REF_CORR, lag1, lag2)
    assert tuple(int(i) for i in max_index[:2]) == (9, 1)
    assert abs(shift_px[0] - 9.33682107) < 2e-2
    assert abs(shift_px[1] - 1.42187891) < 2e-2
    assert abs(shift_arcsec[0] - (15 + shift_px[0])) < 1e-9


def test_oracle_parallel_fanout_equals_serial():
    """counts > 1 (np.array_split chunks over processes, shared-memory images) == in-process loop."""
    from tests import helpers as H
    small, hs, large, hl, _ = H.scene(small_n=48, large_n=64)
    lags = (17.0 + 2.0 * (np.arange(4) - 2), -9.0 + 2.0 * (np.arange(3) - 1), None, None, [0.0, 0.3])
    a = H.oracle_carrington(small, hs, large, hl, lags, (40, 36))
    b = H.oracle_carrington(small, hs, large, hl, lags, (40, 36), counts=3)
    assert np.array_equal(a, b, equal_nan=True)
    a = H.oracle_helio(small, hs, large, hl, lags)
    b = H.oracle_helio(small, hs, large, hl, lags, counts=2)
    assert np.array_equal(a, b, equal_nan=True)


def test_cfg1_golden_fixture_is_reproducible():
    """The committed cfg1 fixture regenerates from its seed: same scene checksums and the oracle reproduces three of its
    lag-points (a full regeneration takes ~40 s: tests/golden/make_golden_cfg1.py)."""
    import os
    from tests.conftest import GOLDEN
    from tests.golden import make_golden_cfg1 as G
    from tests import helpers as H
    g = np.load(os.path.join(GOLDEN, "cfg1_corr.npz"))
    small, hs, large, hl, truth = G.scene()
    assert abs(np.nansum(small) - float(g["small_sum"])) <= 1e-9 * abs(float(g["small_sum"]))
    lags = G.lags(truth)
    sub = (lags[0][4:6], lags[1][5:6], None, None, None)
    got = H.oracle_helio(small, hs, large, hl, sub, parallelism=True)
    assert np.abs(got[:, 0, 0, 0, 0, 0] - g["parallel"][4:6, 5, 0, 0, 0, 0]).max() <= 1e-12
    am = np.unravel_index(np.nanargmax(g["serial"]), g["serial"].shape)
    assert (lags[0][am[0]], lags[1][am[1]]) == (truth["lag_crval1"], truth["lag_crval2"])


@pytest.mark.parametrize("name", ["equator", "north", "south", "rolled", "high", "lonpole_ok", "lonpole_180"])
def test_car_wcs_matches_wcslib(car_golden, name):
    """Plate-carree headers of the align_using_initial_carrington path, oblique cases included (CRVAL2 != 0)."""
    from tests.conftest import car_header
    g = car_golden
    w = O.CarWCS(car_header(g, name))
    lon, lat = w.pixel_to_world(g[name + "/px"], g[name + "/py"])
    assert np.abs(lon - g[name + "/lon"]).max() <= 1e-11   # raw wcslib longitude branch included
    assert np.abs(lat - g[name + "/lat"]).max() <= 1e-11
    bx, by = w.world_to_pixel(g[name + "/lon"], g[name + "/lat"])
    assert np.abs(bx - g[name + "/back_x"]).max() <= 1e-9
    assert np.abs(by - g[name + "/back_y"]).max() <= 1e-9
    assert abs(w.latp - float(g[name + "/latpole_used"])) <= 1e-12


def test_car_wcs_invalid_pole_raises_like_wcslib(car_golden):
    from tests.conftest import car_header
    assert [str(n) for n in car_golden["invalid"]] == ["lonpole_bad"]
    with pytest.raises(O.InvalidTransformError):
        O.CarWCS(car_header(car_golden, "lonpole_bad"))


@pytest.mark.parametrize("tag", ["lag_ns", "lag_nn", "lag_roll", "lag_cdelt"])
def test_car_composite_matches_wcslib(car_golden, tag):
    from tests.conftest import car_header
    g = car_golden
    wa, wb = O.CarWCS(car_header(g, tag + "/A")), O.CarWCS(car_header(g, tag + "/B"))
    lon, lat = wa.pixel_to_world(g[tag + "/gx"], g[tag + "/gy"])
    x, y = wb.world_to_pixel(lon, lat)
    assert np.abs(x - g[tag + "/x"]).max() <= 1e-9
    assert np.abs(y - g[tag + "/y"]).max() <= 1e-9


BORDER_CASES = ["hri2048", "px50", "hri512", "spice", "far"]


def _border_header(g, name):
    h = dict(zip([str(k) for k in g[name + "/keys"]], [float(v) for v in g[name + "/vals"]]))
    h["CUNIT1"] = h["CUNIT2"] = str(g[name + "/unit"])
    h["NAXIS1"], h["NAXIS2"] = int(h["NAXIS1"]), int(h["NAXIS2"])
    h["CTYPE1"], h["CTYPE2"] = "HPLN-TAN", "HPLT-TAN"
    return h


@pytest.mark.parametrize("name", BORDER_CASES)
def test_wcslib_restatement_is_bit_exact_on_border_pixels(name):
    """The zero lag of the sub-map path: pixel -> sky -> ang2pipi -> pixel with IDENTICAL headers is the identity up to
    wcslib's rounding noise, whose sign decides the border pixels.  The oracle's scalar restatement of wcslib reproduces
    astropy 4.3.1 / wcslib 7.6 BIT FOR BIT for every border pixel (sky coordinates, round-trip pixels, keep/drop)."""
    import os
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "border_golden.npz"))
    h = _border_header(g, name)
    w = O.WcslibTan.from_header(h)
    bx, by = g[name + "/bx"], g[name + "/by"]
    step = 1 if bx.size < 3000 else 3  # every border pixel, or every third one of the 2048^2 image (pure-Python loop)
    for k in range(0, bx.size, step):
        lng, lat = w.p2s(float(bx[k]), float(by[k]))
        assert lng == g[name + "/lon"][k] and lat == g[name + "/lat"][k], (name, k)
        l2, b2 = float(O.ang2pipi(np.float64(lng))), float(O.ang2pipi(np.float64(lat)))
        assert l2 == g[name + "/lon_pipi"][k] and b2 == g[name + "/lat_pipi"][k]
        x, y = w.s2p(l2, b2)
        assert x == g[name + "/rx"][k] and y == g[name + "/ry"][k], (name, k)
    if not w.unity:
        assert np.array_equal(np.array(w.imgpix), g[name + "/imgpix"])  # lin.c matinv restated
    # about half of the border pixels fall to the bounds rule
    n = int(g[name + "/dropped"].sum())
    assert 0.4 * bx.size < n < 0.6 * bx.size


@pytest.fixture(scope="module")
def wcstan_c():
    """oracle/_build/liboracle_wcstan.so, built by oracle/Makefile (gcc; __graft_entry__.build() runs it too)."""
    import os
    import subprocess
    from tests.conftest import ROOT
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    O._WCSTAN_LIB = None
    assert O._wcstan_lib() is not None
    return O


@pytest.mark.parametrize("name", BORDER_CASES)
def test_wcslib_c_twin_is_bit_exact_on_border_pixels(wcstan_c, name):
    """oracle/csrc/wcslib_tan.c -- the oracle's wcslib restatement in C, which lets it re-evaluate every pixel of a
    2048 x 2048 grid -- against astropy 4.3.1 / wcslib 7.6 on EVERY border pixel of the golden headers, bit for bit."""
    import os
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "border_golden.npz"))
    h = _border_header(g, name)
    x, y, lng, lat = wcstan_c.wcslib_pixel_to_pixel(h, h, g[name + "/bx"], g[name + "/by"])
    assert np.array_equal(lng, g[name + "/lon"]) and np.array_equal(lat, g[name + "/lat"])
    assert np.array_equal(x, g[name + "/rx"]) and np.array_equal(y, g[name + "/ry"])


def test_wcslib_c_twin_equals_python_class_on_interior_pixels_and_lags(wcstan_c):
    """... and against the Python class on what the golden file does not hold: interior pixels, non-identical header
    pairs (CRVAL / CDELT / CROTA lags), far off-axis pixels, headers in degrees."""
    import os
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "border_golden.npz"))
    rng = np.random.default_rng(3)
    for name in BORDER_CASES:
        h = _border_header(g, name)
        nx, ny = h["NAXIS1"], h["NAXIS2"]
        px = np.concatenate([rng.integers(0, nx, 150).astype(float), rng.uniform(-0.3 * nx, 1.3 * nx, 50)])
        py = np.concatenate([rng.integers(0, ny, 150).astype(float), rng.uniform(-0.3 * ny, 1.3 * ny, 50)])
        for d in [dict(), dict(CRVAL1=h["CRVAL1"] + 3.7 * h["CDELT1"]), dict(CRVAL2=h["CRVAL2"] - 11.0 * h["CDELT2"]),
                  dict(CDELT1=h["CDELT1"] * 1.0003)]:
            h2 = dict(h, **d)
            got = wcstan_c.wcslib_pixel_to_pixel(h, h2, px, py)
            want = wcstan_c.wcslib_pixel_to_pixel(h, h2, px, py, force_python=True)
            for a, b in zip(got, want):
                assert np.array_equal(a, b, equal_nan=True), (name, d)


def test_oracle_zero_lag_uses_wcslib_border_decision():
    import os
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "border_golden.npz"))
    h = _border_header(g, "px50")
    x, y = O.extract_coordinates_pixels(h, dict(h))
    bx, by = g["px50/bx"].astype(int), g["px50/by"].astype(int)
    assert np.array_equal(x[by, bx], g["px50/rx"]) and np.array_equal(y[by, bx], g["px50/ry"])
    drop = (x < 0) | (x > 49) | (y < 0) | (y > 49)
    assert int(drop.sum()) == int(g["px50/dropped"].sum()) and not drop[1:-1, 1:-1].any()
