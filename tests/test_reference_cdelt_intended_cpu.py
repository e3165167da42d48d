"""The DEFAULT CDELT-lag semantics ("intended": CDELTi += d, PC rebuilt with the new lambda) against the reference's own
code: `AlignCommonUtil.correct_pointing_header` (Util.py:161-215) applied to the header first, then the reference's
`Alignment` at its zero lag / sweeping its working CRVAL and CROTA lags (tests/golden/make_golden_cdelt_intended.py).

CPU half: (1) the library's `coreg_shift_header` and the oracle's `shift_header` give the cards the reference's function
gives; (2) the oracle's sweep reproduces the reference's coefficients, Carrington frame and both helioprojective
semantics.  The HIP path against the same entries: tests/test_gpu_reference_cdelt_intended.py."""
import numpy as np
import pytest

from tests import cdelt_cases as K


def _unit_lags(hs, d):
    """arcsec lags -> the header's unit, as `Alignment._set_initial_header_values` converts them (alignment.py:819-837)."""
    from euispice_coreg_amd.utils import header as hdrutil
    unit = hs["CUNIT1"]
    if "arcsec" in unit:
        return tuple(d)
    c = [float(hdrutil.convert(hdrutil.ang2pipi(np.asarray([v]), "arcsec"), "arcsec", unit)[0]) for v in d[:4]]
    return (c[0], c[1], c[2], c[3], d[4])


@pytest.mark.parametrize("name", K.scene_names())
def test_shift_header_gives_the_cards_of_correct_pointing_header(name):
    """Host ABI (`coreg_shift_header`, csrc/geometry.hpp) and oracle (`shift_header`) against Util.py:161-215, cards in
    memory (before astropy rounds them to 16 digits in the file)."""
    from euispice_coreg_amd import _lib
    from oracle import coreg_oracle as O
    from tests import helpers as H
    small, hs, large, hl, _, sc = K.scene(name)
    n = 0
    for c in K.corrected(name):
        d = _unit_lags(hs, c["lag"])
        st = H.oracle_state(small, hs, large, hl, ([0.0], [0.0], None, None, None), unit_lag=hs["CUNIT1"])
        O.set_initial_header_values(st)
        want = c["memory"]
        rc, got = _lib.shift_header(st.hdr_small, *d)
        assert rc == 0
        orc = dict(st.hdr_small)
        O.shift_header(st, orc, *d)
        # a header in degrees: the sweep driver passes the lags through ang2pipi (alignment.py:819-837 -> Util.py:76-80:
        # 180 deg is added and taken off again), which leaves them exact to an ulp of 180 deg = 2.8e-14 deg only --
        # correct_pointing_header converts them directly.  1.3e-11 of a CDELT of 1e-3 deg, i.e. 1e-9 px across the image
        loose = "arcsec" not in hs["CUNIT1"]
        for k in K.CARDS:
            tol = 4e-16 * max(abs(want[k]), 1e-3 if k.startswith("PC") else 0.0)
            if loose:
                tol = 6e-14 if k[:5] in ("CRVAL", "CDELT") else (1e-10 if k.startswith("PC") else tol)
            assert abs(getattr(got, k.lower()) - want[k]) <= tol, (k, c["index"], getattr(got, k.lower()), want[k])
            assert abs(orc[k] - want[k]) <= tol, (k, c["index"], orc[k], want[k])
        # what the reference then read from the file: astropy 4.3.1 writes 16 significant digits in at most 20 characters
        # (0.00029666666666666665 -> "0.000296666666666666"), 2.2e-15 of the value at worst here
        assert all(abs(c["read"][k] - want[k]) <= 4e-15 * abs(want[k]) for k in K.CARDS)
        n += 1
    assert n >= 8


def _oracle_map(name, frame, index):
    from oracle import coreg_oracle as O
    from tests import helpers as H
    small, hs, large, hl, sub, sc = K.scene(name)
    ax = K.lag_axes(sc)
    lags = (ax[0], ax[1], ax[2], ax[3], ax[4])
    carr = sc["carrington"]
    st = H.oracle_state(small.astype(np.float64), hs, large.astype(np.float64), hl, lags, order=sc["order"],
                        shape=carr["shape"], lonlims=carr["lonlims"], latlims=carr["latlims"])
    flat = np.ravel_multi_index(tuple(index.T), tuple(len(a) for a in ax))
    full = O.find_best_header_parameters(st, "carrington" if frame == "carrington" else "helioprojective",
                                         parallelism=(frame == "helio_parallel"), lag_subset=np.unique(flat))
    return full[..., 0][tuple(index.T)]


@pytest.mark.parametrize("name,frame", [(n, f) for n in K.scene_names() for f in K.frames(n)])
def test_oracle_reproduces_the_reference_under_the_intended_semantics(name, frame):
    index, want, mode = K.entries(name, frame)
    assert len(want) >= 40 and "zero_lag" in mode and "swept" in mode
    got = _oracle_map(name, frame, index)
    d = np.abs(got - want)
    print(f"{name} {frame}: {len(want)} reference entries ({mode.count('zero_lag')} zero-lag, {mode.count('swept')} "
          f"swept), {int((index[:, 2:4] != 1).any(axis=1).sum())} with a CDELT lag: max |oracle - reference| = {d.max():.2e}")
    assert d.max() <= 1e-9, (index[np.argmax(d)], d.max())


def test_the_fixture_covers_what_was_asked():
    """>= 20 (cdelt1, cdelt2, crota, crval) combinations, unequal CDELT and a header in degrees included."""
    _, m = K.load()
    combos = {(e["scene"],) + tuple(e["index"]) for e in m["entries"]}
    with_cdelt = {c for c in combos if c[3] != 1 or c[4] != 1}
    assert len(with_cdelt) >= 150
    assert m["scenes"]["E"]["hdr_small"]["CUNIT1"] == "deg"
    assert m["scenes"]["E"]["hdr_small"]["CDELT1"] != m["scenes"]["E"]["hdr_small"]["CDELT2"]
    assert m["scenes"]["U"]["hdr_small"]["CDELT1"] != m["scenes"]["U"]["hdr_small"]["CDELT2"]
    # the lags do something: a CDELT lag moves the coefficient by far more than the tolerance
    index, want, _ = K.entries("A", "carrington")
    base = {tuple(i[[0, 1, 4]]): w for i, w in zip(index, want)}
    assert max(abs(w - v) for i, w in zip(index, want) for v in [base[tuple(i[[0, 1, 4]])]]) > 1e-4


def test_cfg5_at_4096_the_oracle_equals_the_references_own_code_on_the_cdelt_planes():
    """BASELINE configs[4] at its stated size: 38 lag-points of the 41 x 41 x 5 x 5 x 11 map from the reference's own
    `Alignment` on the 4096^2 grid (headers through `correct_pointing_header`; make_golden_cdelt_intended.py cfg5), and the
    oracle's values at the same lag-points as make_golden_cfg5_sample.py computed them in the build container (seconds per
    lag-point at this size: committed, not recomputed here).  The HIP path against both: test_gpu_fullsize.py."""
    import os
    ref = np.load(os.path.join(K.GOLDEN, "cdelt_intended_cfg5.npz"))
    ora = np.load(os.path.join(K.GOLDEN, "cfg5_sample.npz"))
    dims = (41, 41, 5, 5, 11)
    assert np.array_equal(ref["fingerprint"], ora["fingerprint"])
    assert np.array_equal(np.ravel_multi_index(tuple(ref["index"].T), dims), ora["reference_index"])
    ri = ref["index"]
    with_cdelt = (ri[:, 2] != 2) | (ri[:, 3] != 2)
    assert with_cdelt.sum() >= 16 and len({tuple(v) for v in ri[:, 2:4]}) >= 8 and (ref["mode"] == 0).sum() >= 4
    d = np.abs(ora["oracle_at_reference"] - ref["corr"])
    print("cfg5:", ri.shape[0], "lag-points of the reference's own code,", int(with_cdelt.sum()),
          "with a CDELT lag: max |oracle - reference| =", d.max())
    assert d.max() <= 1e-10
    # the header cards the reference's function gave for those planes, against the library's shift_header
    from euispice_coreg_amd import _lib, synthetic
    hs = synthetic.make_scene(small_n=8, large_n=8, n_blobs=1)[1]   # the header does not depend on the size ...
    hs.update(synthetic._header(2048, 2048, 1024.5, 1024.5, hs["CRVAL1"], hs["CRVAL2"], 0.492, 0.492, 3.0))  # ... but CRPIX / CDELT do
    l1 = np.arange(-20.0, 21.0, 1.0)
    lc = np.round(np.arange(-2, 3) * 0.01, 10)
    lr = np.round(np.arange(-5, 6) * 0.1, 10)
    zero = ri[ref["mode"] == 0]
    cards = ref["cards"][-zero.shape[0]:, :9]      # in memory, after correct_pointing_header
    for idx, want in zip(zero, cards):
        rc, got = _lib.shift_header(hs, l1[idx[0]], l1[idx[1]], lc[idx[2]], lc[idx[3]], lr[idx[4]])
        assert rc == 0
        for k, w in zip(K.CARDS, want):
            assert abs(getattr(got, k.lower()) - w) <= 4e-16 * max(abs(w), 1e-3), (k, idx, getattr(got, k.lower()), w)


@pytest.mark.parametrize("name", K.scene_names())
def test_products_correct_pointing_header_gives_the_references_cards(name):
    """`euispice_coreg_amd.utils.header.correct_pointing_header` (what `AlignmentResults.write_corrected_fits`
    / `return_corrected_header` apply) against the cards the REFERENCE's function produced from the same header and lags
    (Util.py:161-215), CDELT lags included -- 30 headers over three scenes, in memory (before astropy rounds to 16 digits)."""
    from euispice_coreg_amd.utils import header as hdrutil
    _, hs, _, _, _, _ = K.scene(name)
    n = 0
    for c in K.corrected(name):
        h = dict(hs)
        a = [None if v == 0.0 else float(v) for v in c["lag"]]   # as the generator handed them to the reference
        hdrutil.correct_pointing_header(h, lag_cdelt1=a[2], lag_cdelt2=a[3], lag_crota=a[4], lag_crval1=a[0], lag_crval2=a[1])
        for k in K.CARDS:
            w = c["memory"][k]
            assert abs(h[k] - w) <= 4e-16 * max(abs(w), 1e-3 if k.startswith("PC") else 0.0), (k, c["index"], h[k], w)
        n += 1
    assert n >= 8
