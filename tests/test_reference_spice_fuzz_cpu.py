"""CPU tests of the SPICE caller against the RANDOM family the reference's own `AlignmentSpice` produced
(tests/golden/make_golden_spice_fuzz.py -> tests/golden/spice_fuzz_golden.{npz,json}: 8 seeded windows -- raster size,
detector, NBIN2, PXBEG2, PC4_1, NaN voxels, fully-NaN spectra -- and random call options):

  * `AlignmentSpice._extract_spice_data_header` (alignment_spice.py:189-323, Util.py:431-455): the collapsed image
    bit for bit (np.nansum of a fully-NaN spectrum is 0.0, a VALID sample: reproduced), the 2-D header card for card
    (`extend_pixel_size`: CDELT1 after _correct_solar_rotation, alignment_spice.py:223-248);
  * the oracle's sweep on the reference's prepared image / header: the reference's correlation map.
GPU half: tests/test_gpu_reference_spice_fuzz.py."""
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN


def load():
    g = np.load(os.path.join(GOLDEN, "spice_fuzz_golden.npz"))
    with open(os.path.join(GOLDEN, "spice_fuzz_golden.json")) as f:
        return g, json.load(f)


def scenes():
    with open(os.path.join(GOLDEN, "spice_fuzz_golden.json")) as f:
        return sorted(json.load(f)["scenes"])


def inputs(g, m, name):
    """(cube float32 [1, nw, ny, nx], 4-D header, reference image float32, its header) as the reference read them."""
    cube = (g[f"{name}/image"][None, :, :] * g[f"{name}/profile"][:, None, None])[None].astype(np.float32)
    for w, y, x in g[f"{name}/nan_voxels"]:
        cube[0, w, y, x] = np.nan
    for y, x in g[f"{name}/nan_spectra"]:
        cube[0, :, y, x] = np.nan
    c = m["scenes"][name]
    return cube, dict(c["hdr4d"]), g[f"{name}/large"], dict(c["hdr_large"])


def make_spice(name, g, m, small=None, large=None, **extra):
    from euispice_coreg_amd.hdrshift import AlignmentSpice
    c = m["scenes"][name]
    cube, h4, lg, hl = inputs(g, m, name)
    kw = dict(c["ctor"])
    kw.pop("counts_cpu_max", None)
    kw.update({k: list(v) for k, v in (c.get("quantities") or {}).items()})  # plain numbers: angstrom / arcsec
    kw.update(extra)
    A = AlignmentSpice(large if large is not None else (lg, hl), small if small is not None else (cube, h4), level=2,
                       small_fov_window=0, cdelt_semantics="reference", **kw)
    return A, c


def test_fixture_is_what_the_generator_describes():
    g, m = load()
    sc = m["scenes"]
    assert m["interpreter"]["astropy"] == "4.3.1" and m["interpreter"]["seed"] == 88000 and len(sc) == 8
    assert {c["hdr4d"]["DETECTOR"] for c in sc.values()} == {"SW", "LW"}
    assert {c["hdr4d"]["NBIN2"] for c in sc.values()} == {2, 4}
    for opt in ("cut_from_center", "extend_pixel_size"):
        assert any(opt in c["call_kwargs"] for c in sc.values()) and any(opt not in c["call_kwargs"] for c in sc.values())
    for q in ("wavelength_interval_to_sum", "sub_fov_window"):
        assert any(q in c["quantities"] for c in sc.values()) and any(q not in c["quantities"] for c in sc.values())
    assert any(c["ctor"]["parallelism"] for c in sc.values())
    assert any(c.get("prepared_zero", 0) > 0 for c in sc.values())  # fully-NaN spectra that became 0.0


@pytest.mark.parametrize("name", scenes())
def test_spice_preparation_equals_the_reference(name):
    g, m = load()
    A, c = make_spice(name, g, m)
    A.hdr_large = dict(c["hdr_large"])
    A.extend_pixel_size = bool(c["call_kwargs"].get("extend_pixel_size", False))
    A.cut_from_center = c["call_kwargs"].get("cut_from_center")
    A._extract_spice_data_header(level=2)
    for k, v in c["hdr_small"].items():
        if k in ("WCSAXES", "LATPOLE", "MJDREF", "DATEREF", "TIMESYS", "MJD-OBS", "DATE-OBS", "RSUN_REF") and k not in A.hdr_small:
            continue  # bookkeeping cards wcslib adds; not read by the sweep
        assert k in A.hdr_small, k
        if isinstance(v, float):
            assert A.hdr_small[k] == pytest.approx(v, rel=1e-15, abs=1e-300), (k, A.hdr_small[k], v)
        else:
            assert A.hdr_small[k] == v, (k, A.hdr_small[k], v)
    if f"{name}/data_small" in g.files:
        want = g[f"{name}/data_small"]
        assert A.data_small.shape == want.shape
        assert np.array_equal(A.data_small, want, equal_nan=True)


@pytest.mark.parametrize("name", [n for n in scenes()])
def test_oracle_sweep_on_the_references_prepared_spice_image(name):
    from oracle import coreg_oracle as O
    g, m = load()
    c = m["scenes"][name]
    if f"{name}/data_small" not in g.files:
        pytest.skip("parallel branch: the reference deletes its prepared image (GPU test covers the case end to end)")
    hs, hl = dict(c["hdr_small"]), dict(c["hdr_large"])
    O.check_and_create_pcij_matrix(hl)
    ctor = c["ctor"]
    st = O.SweepState(hs, hl, g[f"{name}/data_small"], g[f"{name}/large"].astype(np.float64), ctor["lag_crval1"],
                      ctor["lag_crval2"], None, None, ctor["lag_crota"], unit_lag="arcsec", cdelt_semantics="reference")
    got = O.find_best_header_parameters(st, "helioprojective", parallelism=False)
    want = g[f"{name}/corr"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    # samples are rounded to float32 (quirk Q7): a 1e-10 px difference between the oracle's TAN restatement and wcslib
    # flips a rounding now and then; on a 7 000-sample raster one flip is worth 3e-10 (measured: 0.0 on six windows,
    # 2.4e-12 / 3.2e-10 on two lag-points of P05)
    assert np.nanmax(np.abs(got - want)) <= 1.1e-9, np.nanmax(np.abs(got - want))
    assert np.nanargmax(got) == np.nanargmax(want)
