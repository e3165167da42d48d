"""
CPU tests of the C ABI that need no GPU: the shared library loads, exports every symbol include/coreg_hip.h
declares, and its host-side header arithmetic (header shifting, TAN->TAN homography, Carrington origin) agrees
with the oracle.  No compute entry point is called here.
"""
import os
import re

import numpy as np
import pytest

from oracle import coreg_oracle as O
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()  # hipcc cross-compiles without a GPU; no-op when the .so is newer than its sources
    from euispice_coreg_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "coreg_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(coreg_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    cdll = lib.load_library()
    bound = {name for name, _, _ in lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(cdll, name), name
    assert b"gfx950" in cdll.coreg_version()


def test_struct_layouts_match_header(lib):
    import ctypes as C
    assert C.sizeof(lib.Wcs2d) == 8 + 17 * 8 + 8  # 2 x int32, 17 doubles, proj + reserved
    assert C.sizeof(lib.Lags) == 5 * 16
    assert C.sizeof(lib.CarrGrid) == 6 * 8 + 2 * 8
    assert C.sizeof(lib.Stats) == 3 * 8 + 4 * 8 + 2 * 4


def test_missing_library_is_loud(monkeypatch, tmp_path, lib):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        lib.load_library()


def _state():
    small, hs, large, hl, _ = H.scene(small_n=48, large_n=64)
    st = H.oracle_state(small, hs, large, hl, ([0.0], [0.0], None, None, None))
    O.set_initial_header_values(st)
    return st


@pytest.mark.parametrize("d", [(24.0, 6.0, 0.0, 0.0, 0.0), (-3.0, 11.5, 0.0, 0.0, 0.75), (1.0, 2.0, 0.02, 0.0, 0.0),
                               (1.0, 2.0, 0.0, -0.03, 0.0), (0.0, 0.0, 0.01, -0.02, -0.4)])
def test_shift_header_matches_oracle(lib, d):
    """coreg_shift_header == alignment.py:401-468 restated in the oracle (intended CDELT semantics), including the
    'PC rebuilt only when a crota/cdelt lag is non-zero' rule (quirk Q3)."""
    st = _state()
    st.hdr_small["PC1_2"] *= 1.0000001  # make the header PC slightly inconsistent with CROTA: Q3 becomes visible
    want = dict(st.hdr_small)
    O.shift_header(st, want, *d)
    rc, got = lib.shift_header(st.hdr_small, *d)
    assert rc == 0
    for k, f in [("CRVAL1", "crval1"), ("CRVAL2", "crval2"), ("CDELT1", "cdelt1"), ("CDELT2", "cdelt2"),
                 ("PC1_1", "pc1_1"), ("PC1_2", "pc1_2"), ("PC2_1", "pc2_1"), ("PC2_2", "pc2_2"), ("CROTA", "crota")]:
        assert getattr(got, f) == pytest.approx(want[k], rel=0, abs=1e-15 * max(1.0, abs(want[k]))), k


def test_shift_header_reference_cdelt_semantics(lib):
    st = _state()
    rc, got = lib.shift_header(st.hdr_small, 1.0, 2.0, 0.02, 0.0, 0.0, cdelt_semantics=lib.CDELT_REFERENCE)
    assert rc == 0 and got.cdelt1 == st.hdr_small["CDELT1"]  # d_cdelt1 is never written (alignment.py:423-430)
    rc, _ = lib.shift_header(st.hdr_small, 1.0, 2.0, 0.0, 0.01, 0.0, cdelt_semantics=lib.CDELT_REFERENCE)
    assert rc == 1  # the reference's worker dies (alignment.py:440)


@pytest.mark.parametrize("d", [(24.0, 6.0, 0.0, 0.0, 0.75), (-30.0, 30.0, 0.0, 0.0, 0.0), (5.0, -5.0, 0.01, -0.02, -1.0)])
def test_homography_matches_oracle_tan(lib, d):
    """coreg_homography(hdr_target -> shifted hdr) reproduces the per-pixel spherical-trig route
    WCS.world_to_pixel(ang2pipi(WCS.pixel_to_world())) (alignment.py:1038-1069) to < 1e-9 px."""
    st = _state()
    hdr = dict(st.hdr_small)
    O.shift_header(st, hdr, *d)
    x, y = O.extract_coordinates_pixels(st.hdr_small, hdr)
    hm = lib.homography(st.hdr_small, hdr)
    jj, ii = np.mgrid[0:st.hdr_small["NAXIS2"], 0:st.hdr_small["NAXIS1"]].astype(float)
    w = hm[2, 0] * ii + hm[2, 1] * jj + hm[2, 2]
    assert np.abs((hm[0, 0] * ii + hm[0, 1] * jj + hm[0, 2]) / w - x).max() < 1e-9
    assert np.abs((hm[1, 0] * ii + hm[1, 1] * jj + hm[1, 2]) / w - y).max() < 1e-9
    # small grid -> large header (the once-only sub-map, alignment.py:993)
    x, y = O.extract_coordinates_pixels(st.hdr_small, st.hdr_large)
    hm = lib.homography(st.hdr_small, st.hdr_large)
    w = hm[2, 0] * ii + hm[2, 1] * jj + hm[2, 2]
    assert np.abs((hm[0, 0] * ii + hm[0, 1] * jj + hm[0, 2]) / w - x).max() < 1e-9


def test_homography_matches_wcslib_golden(lib, wcs_golden):
    from tests.conftest import golden_header
    g = wcs_golden
    for tag in ["lag_hri", "sub_hri_fsi", "lag_spice", "lag_cdelt"]:
        ha, hb = golden_header(g, tag + "/A"), golden_header(g, tag + "/B")
        ha.setdefault("CROTA", 0.0)
        hb.setdefault("CROTA", 0.0)
        hm = lib.homography(ha, hb)
        gx, gy = g[tag + "/gx"], g[tag + "/gy"]
        w = hm[2, 0] * gx + hm[2, 1] * gy + hm[2, 2]
        assert np.abs((hm[0, 0] * gx + hm[0, 1] * gy + hm[0, 2]) / w - g[tag + "/x"]).max() < 1e-9, tag
        assert np.abs((hm[1, 0] * gx + hm[1, 1] * gy + hm[1, 2]) / w - g[tag + "/y"]).max() < 1e-9, tag


def test_lag_homography_family_matches_direct(lib):
    """The factored per-lag map used by the sweep (B * P[i2] * Q[i1]) == the direct composition == the oracle."""
    st = _state()
    lags = lib.LagSet(np.arange(-30, 31, 6.0), np.arange(-30, 31, 10.0), [0.0, 0.01], [-0.02, 0.0], [-0.5, 0.0, 0.75])
    jj, ii = np.mgrid[0:st.hdr_small["NAXIS2"], 0:st.hdr_small["NAXIS1"]].astype(float)
    for idx in [(0, 0, 0, 0, 0), (10, 6, 1, 0, 2), (5, 3, 0, 1, 1), (7, 2, 1, 1, 0)]:
        d = [lags.arrays[k][idx[k]] for k in range(5)]
        hdr = dict(st.hdr_small)
        O.shift_header(st, hdr, d[0], d[1], d[2], d[3], d[4])
        rc, hm = lib.lag_homography(st.hdr_small, st.hdr_small, lags, idx)
        assert rc == 0
        direct = lib.homography(st.hdr_small, hdr)
        w = hm[2, 0] * ii + hm[2, 1] * jj + hm[2, 2]
        wd = direct[2, 0] * ii + direct[2, 1] * jj + direct[2, 2]
        xf, xd = (hm[0, 0] * ii + hm[0, 1] * jj + hm[0, 2]) / w, (direct[0, 0] * ii + direct[0, 1] * jj + direct[0, 2]) / wd
        yf, yd = (hm[1, 0] * ii + hm[1, 1] * jj + hm[1, 2]) / w, (direct[1, 0] * ii + direct[1, 1] * jj + direct[1, 2]) / wd
        assert np.abs(xf - xd).max() < 1e-9 and np.abs(yf - yd).max() < 1e-9, idx
        x, y = O.extract_coordinates_pixels(st.hdr_small, hdr)
        assert np.abs(xf - x).max() < 1e-9 and np.abs(yf - y).max() < 1e-9, idx


def test_carrington_origin_matches_oracle(lib):
    st = _state()
    hdr = dict(st.hdr_small)
    O.shift_header(st, hdr, 12.0, -7.0, 0.0, 0.0, 0.5)
    p = O.carrington_params(hdr, 1.004)
    x0, y0 = lib.carrington_origin(hdr)
    assert x0 == pytest.approx(p["x0"], abs=1e-11) and y0 == pytest.approx(p["y0"], abs=1e-11)


@pytest.mark.parametrize("tag", ["lag_ns", "lag_nn", "lag_roll", "lag_cdelt"])
def test_car_map_host_matches_wcslib(car_golden, tag):
    """geometry.hpp: celset for plate-carree headers + the sphere rotation between two CAR maps (host helper
    coreg_car_map) against astropy / wcslib composites."""
    from euispice_coreg_amd import _lib
    from tests.conftest import car_header
    g = car_golden
    out = _lib.car_map(car_header(g, tag + "/A"), car_header(g, tag + "/B"), g[tag + "/gx"], g[tag + "/gy"])
    assert out is not None
    assert np.abs(out[0] - g[tag + "/x"]).max() <= 1e-9
    assert np.abs(out[1] - g[tag + "/y"]).max() <= 1e-9


def test_car_map_host_reports_invalid_pole(car_golden):
    from euispice_coreg_amd import _lib
    from tests.conftest import car_header
    g = car_golden
    assert _lib.car_map(car_header(g, "equator"), car_header(g, "lonpole_bad"), [1.0], [2.0]) is None
    with pytest.raises(_lib.CoregError):  # TAN header where a CAR one is required
        _lib.car_map(car_header(g, "equator"), dict(car_header(g, "equator"), CTYPE1="HPLN-TAN"), [1.0], [2.0])


@pytest.mark.parametrize("name", ["hri2048", "px50", "hri512", "spice", "far"])
def test_library_wcslib_chain_is_bit_exact_on_border_pixels(name):
    """csrc/geometry.hpp WcslibTan (the host code that decides the border pixels of the zero lag) against the same
    astropy 4.3.1 / wcslib 7.6 vectors as the oracle: bit for bit, every border pixel."""
    import os
    from tests.conftest import GOLDEN
    from tests.test_oracle_golden import _border_header
    from euispice_coreg_amd import _lib
    g = np.load(os.path.join(GOLDEN, "border_golden.npz"))
    h = _border_header(g, name)
    x, y, lon, lat = _lib.wcslib_pixel_to_pixel(h, h, g[name + "/bx"], g[name + "/by"])
    assert np.array_equal(lon, g[name + "/lon"]) and np.array_equal(lat, g[name + "/lat"])
    assert np.array_equal(x, g[name + "/rx"]) and np.array_equal(y, g[name + "/ry"])
    nx, ny = h["NAXIS1"], h["NAXIS2"]
    drop = ~((x >= 0) & (x <= nx - 1) & (y >= 0) & (y <= ny - 1))
    assert np.array_equal(drop, g[name + "/dropped"])


def test_hand_issued_lds_reads_are_not_touched_before_their_wait():
    """kernels.hpp Taps<N>: reads and s_waitcnt sit in separate inline-asm statements; the disassembly of the built
    library must show no use of a tap register in between (csrc/check_isa.py, also run by build())."""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("ROCm LLVM tools not installed")
    from euispice_coreg_amd import _lib
    chk = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc", "check_isa.py")
    p = subprocess.run([sys.executable, chk, _lib.LIB_PATH], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "hand-issued LDS read groups" in p.stdout
