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


def test_shift_header_says_when_a_lag_leaves_no_header_to_evaluate(lib):
    """CDELT + d_cdelt = 0 (or a non-finite lag): return code 2 -- the sweeps leave such lag-points NaN, as they leave
    those whose worker dies in the reference (astropy refuses a header with CDELT = 0)."""
    st = _state()
    cd1, cd2 = st.hdr_small["CDELT1"], st.hdr_small["CDELT2"]
    assert lib.shift_header(st.hdr_small, 0.0, 0.0, -cd1, 0.0, 0.0)[0] == 2
    assert lib.shift_header(st.hdr_small, 0.0, 0.0, 0.0, -cd2, 0.3)[0] == 2
    assert lib.shift_header(st.hdr_small, 0.0, 0.0, 0.0, 0.0, float("nan"))[0] == 2
    assert lib.shift_header(st.hdr_small, 0.0, 0.0, -0.5 * cd1, 0.0, 0.0)[0] == 0
    # reference semantics never write d_cdelt1 into the header: nothing degenerates
    assert lib.shift_header(st.hdr_small, 0.0, 0.0, -cd1, 0.0, 0.0, cdelt_semantics=lib.CDELT_REFERENCE)[0] == 0


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


def test_library_wcslib_car_chain_is_bit_exact():
    """csrc/geometry.hpp WcslibCar (the host code that decides the border pixels of the identity lag of two Carrington
    maps) against astropy 4.3.1 / wcslib 7.6: nine CAR headers, every border pixel + two diagonals, world coordinates and
    the pixel -> world -> pixel round trip, bit for bit (tests/golden/make_golden_border_car.py)."""
    import os
    from tests.conftest import GOLDEN
    from euispice_coreg_amd import _lib
    g = np.load(os.path.join(GOLDEN, "border_car_golden.npz"))
    names = sorted(set(k.split("/")[0] for k in g.files))
    assert len(names) == 9
    for name in names:
        h = dict(zip([str(k) for k in g[name + "/keys"]], [float(v) for v in g[name + "/vals"]]))
        h.update(CUNIT1=str(g[name + "/unit"]), CUNIT2=str(g[name + "/unit"]), CTYPE1="CRLN-CAR", CTYPE2="CRLT-CAR",
                 NAXIS1=int(h["NAXIS1"]), NAXIS2=int(h["NAXIS2"]))
        x, y, lon, lat = _lib.wcslib_pixel_to_pixel(h, h, g[name + "/bx"], g[name + "/by"])
        assert np.array_equal(lon, g[name + "/lon"]) and np.array_equal(lat, g[name + "/lat"]), name
        assert np.array_equal(x, g[name + "/rx"]) and np.array_equal(y, g[name + "/ry"]), name


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
    assert "hand-issued LDS read groups" in p.stdout and "of 16 reads" in p.stdout  # (the cubic gather too)
    assert "s_load_dwordx8 in k_sweep" in p.stdout


def test_isa_check_catches_a_read_of_a_pending_scalar_load():
    """The checker itself: an SGPR of a hand-issued s_load_dwordx8 named before the lgkmcnt(0) wait is reported, on the
    fall-through path and behind a forward branch; the clean sequence passes."""
    import importlib.util
    from euispice_coreg_amd import _lib
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc",
                                                                          "check_isa.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    head = "0000000000001000 <_ZN5coreg7k_sweepILi0ELi2EfLb0ELb0ELi121EEEvNS_9SweepArgsE>:\n"
    clean = head + "\n".join([
        "\ts_load_dwordx8 s[36:43], s[4:5], 0x0   // 000000001000: C00E0902 00000000",
        "\tv_add_f64 v[4:5], v[0:1], s[44:45]      // 000000001008: D2800004 00000000",
        "\ts_cbranch_execz 2                      // 000000001010: BF880002",
        "\tv_mul_f64 v[6:7], v[4:5], v[4:5]        // 000000001014: D2810006 00000000",
        "\ts_waitcnt lgkmcnt(0)                   // 00000000101C: BF8CC07F",
        "\tv_add_f64 v[4:5], v[0:1], s[36:37]      // 000000001020: D2800004 00000000",
        "\ts_endpgm                               // 000000001028: BF810000"]) + "\n"
    n, bad = chk.check_scalar_loads(clean)
    assert n == 1 and not bad
    early = clean.replace("v[0:1], s[44:45]", "v[0:1], s[38:39]")
    assert chk.check_scalar_loads(early)[1]
    # the branch at 0x1010 skips the wait (target 0x101C + ... = 0x1020 when the offset is 3): the read there is early
    skipping = clean.replace("s_cbranch_execz 2 ", "s_cbranch_execz 3 ")
    assert chk.check_scalar_loads(skipping)[1]


def _car_hdr(nx, ny, crval, cdelt, crpix=None, rot=0.0):
    rho, lam = np.deg2rad(rot), cdelt[1] / cdelt[0]
    crpix = crpix or ((nx + 1) / 2.0, (ny + 1) / 2.0)
    return {"NAXIS": 2, "NAXIS1": nx, "NAXIS2": ny, "CTYPE1": "CRLN-CAR", "CTYPE2": "CRLT-CAR", "CUNIT1": "deg",
            "CUNIT2": "deg", "CRPIX1": crpix[0], "CRPIX2": crpix[1], "CRVAL1": crval[0], "CRVAL2": crval[1],
            "CDELT1": cdelt[0], "CDELT2": cdelt[1], "PC1_1": float(np.cos(rho)), "PC1_2": float(-lam * np.sin(rho)),
            "PC2_1": float(np.sin(rho) / lam), "PC2_2": float(np.cos(rho)), "CROTA": rot}


@pytest.mark.parametrize("tile_w", [8, 32, 128])
def test_car_tile_margin_bounds_the_map(lib, tile_w):
    """k_sweep (MODE_CAR) trusts, for the window it stages and for its 'interior' decision, that the image of a tile of
    the target map lies inside the bounding box of its four mapped corners widened by car_tile_margin.  Checked here
    against the host map for full-Sun target maps (tiles up to the poles), oblique and rolled shifted maps: every pixel
    of every tile must land inside the widened box; polar tiles must get an infinite allowance (global path)."""
    tile_h = 1024 // tile_w
    target = _car_hdr(720, 360, (180.0, 0.0), (0.5, 0.5))
    cases = [_car_hdr(400, 300, (181.3, 0.9), (0.4, 0.4)),
             _car_hdr(300, 340, (170.0, -1.5), (0.3, 0.5), rot=4.0),
             _car_hdr(200, 180, (200.0, 0.6), (0.45, 0.45), crpix=(100.5, -59.0)),  # rows at latitude 30 .. 89.6
             _car_hdr(500, 500, (185.0, 25.0), (0.35, 0.35), rot=-7.0)]              # oblique: poles 25 deg apart
    n_inf = n_checked = 0
    worst = 0.0
    for shifted in cases:
        for j0 in range(0, 360, tile_h):
            for i0 in range(0, 720, max(tile_w, 720 // 24)):
                ii, jj = np.meshgrid(np.arange(i0, min(i0 + tile_w, 720)), np.arange(j0, min(j0 + tile_h, 360)))
                lat = np.abs((jj + 1 - target["CRPIX2"]) * target["CDELT2"])
                m = lib.car_tile_margin(target, shifted, tile_w, float(np.deg2rad(lat.max())))
                assert m is not None and m >= 1.0
                if not np.isfinite(m):
                    n_inf += 1
                    continue
                x, y = lib.car_map(target, shifted, ii.ravel().astype(float), jj.ravel().astype(float))
                cx, cy = lib.car_map(target, shifted, ii[[0, 0, -1, -1], [0, -1, 0, -1]].astype(float),
                                     jj[[0, 0, -1, -1], [0, -1, 0, -1]].astype(float))
                if cx.max() - cx.min() > 0.5 * 360.0 / abs(shifted["CDELT1"]):
                    continue  # the tile straddles the +-180 deg cut of the shifted map: never 'interior', never in LDS
                out = max(cx.min() - x.min(), x.max() - cx.max(), cy.min() - y.min(), y.max() - cy.max())
                worst = max(worst, out / m)
                assert out <= m, (tile_w, shifted["CRVAL2"], i0, j0, out, m)
                n_checked += 1
    print(f"tile_w={tile_w}: {n_checked} tiles checked, {n_inf} with an infinite allowance, worst excess / allowance = {worst:.3f}")
    assert n_checked > 60 and n_inf > 0 and worst > 0.01  # the bound is exercised, not vacuous


def test_multi_plan_matches_the_python_partition(lib):
    """The in-library multi-GPU driver (csrc/multi.hpp) and the one-process-per-GPU path (parallel.py) must cut a lag
    set the same way: mode and block grid for a spread of lag shapes and GPU counts."""
    from euispice_coreg_amd import parallel
    shapes = [(60, 60, 1), (61, 61, 1), (121, 121, 1), (61, 61, 21), (41, 41, 275), (1, 2000, 1), (2000, 1, 1),
              (5, 5, 1), (1, 1, 2001), (3, 3, 275), (7, 2, 40), (2, 7, 40), (16, 16, 1), (1, 1, 1), (9, 9, 2),
              (24, 24, 2), (24, 24, 3), (20, 20, 3), (12, 15, 6), (61, 61, 2), (100, 100, 5), (31, 17, 12), (4, 300, 7)]
    seen = set()
    for n1, n2, inner in shapes:
        for world in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16):
            for per_combo in (True, False):
                mode, gc, g1, g2 = lib.multi_plan(n1, n2, inner, world, per_combo)
                want = parallel.lag_plan((n1, n2, inner, 1, 1), world, per_combo)
                seen.add(mode)
                assert mode == want[0], (n1, n2, inner, world, per_combo, mode, want)
                if mode in ("blocks", "combos"):
                    assert (gc, g1, g2) == want[1:] and gc * g1 * g2 == world, (n1, n2, inner, world, (gc, g1, g2), want)
                    assert (mode == "combos") == (gc > 1)
                else:
                    assert (g1, g2) == parallel.block_grid(n1, n2, world), (n1, n2, inner, world)
    assert seen == {"none", "points", "blocks", "combos", "slices"}
    # cfg4 is a helioprojective sweep (one launch for all 21 CROTA values): blocks of the plane; as a Carrington sweep it
    # would be dealt by combination
    assert lib.multi_plan(61, 61, 21, 8, False)[0] == "blocks" and lib.multi_plan(61, 61, 21, 8, True)[0] == "combos"
