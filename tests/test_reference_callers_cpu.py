"""CPU tests of the callers either side of the sweep against what the REFERENCE's own classes produced
(tests/golden/make_golden_callers.py ran AlignmentSpice, SPICEComposedMapBuilder.process and jitter_correction_imagers in
the build container; tests/golden/callers_golden.{npz,json}):

  * `AlignmentSpice._extract_spice_data_header` (alignment_spice.py:189-323): the collapsed image bit for bit, the 2-D
    header card for card;
  * the oracle's sweep on the reference's prepared image / header: the reference's correlation map;
  * the raster-column -> imager-frame choice of the synthetic-raster builder (map_builder.py:95-99).
The GPU halves (sweeps, raster, session) are in tests/test_gpu_reference_callers.py."""
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN


def load():
    g = np.load(os.path.join(GOLDEN, "callers_golden.npz"))
    with open(os.path.join(GOLDEN, "callers_golden.json")) as f:
        return g, json.load(f)


def spice_inputs(g, m):
    """(cube float32 [1, nw, ny, nx], 4-D header, reference image float32, its header) as the reference read them."""
    cube = (g["spice/image"][None, :, :] * g["spice/profile"][:, None, None])[None].astype(np.float32)
    h4 = dict(m["spice"]["hdr4d"])
    return cube, h4, g["spice/large"], dict(m["spice"]["hdr_large"])


SPICE_CASES = ["helio_serial", "helio_parallel", "helio_interval_cut_subfov", "helio_extend_pixel_size"]


def make_spice(case, g, m, small=None, large=None, **extra):
    from euispice_coreg_amd.hdrshift import AlignmentSpice
    c = m["spice"]["cases"][case]
    cube, h4, lg, hl = spice_inputs(g, m)
    kw = dict(c["ctor"])
    kw.pop("counts_cpu_max", None)
    q = c.get("quantities") or {}
    kw.update({k: list(v) for k, v in q.items()})  # plain numbers: angstrom / arcsec, the units the reference was given
    kw.update(extra)
    A = AlignmentSpice(large if large is not None else (lg, hl), small if small is not None else (cube, h4), level=2,
                       small_fov_window=0, cdelt_semantics="reference", **kw)
    return A, c


@pytest.mark.parametrize("case", SPICE_CASES)
def test_spice_preparation_equals_the_reference(case):
    g, m = load()
    A, c = make_spice(case, g, m)
    A.hdr_large = dict(m["spice"]["hdr_large"])
    A.extend_pixel_size = bool(c["call_kwargs"].get("extend_pixel_size", False))
    A.cut_from_center = c["call_kwargs"].get("cut_from_center")
    A._extract_spice_data_header(level=2)
    ref_hdr = c["hdr_small"]
    for k, v in ref_hdr.items():
        if k in ("WCSAXES", "LATPOLE", "MJDREF", "DATEREF", "TIMESYS", "MJD-OBS", "DATE-OBS", "RSUN_REF") and k not in A.hdr_small:
            continue  # bookkeeping cards wcslib adds; not read by the sweep
        assert k in A.hdr_small, k
        if isinstance(v, float):
            assert A.hdr_small[k] == pytest.approx(v, rel=1e-15, abs=1e-300), (k, A.hdr_small[k], v)
        else:
            assert A.hdr_small[k] == v, (k, A.hdr_small[k], v)
    if f"spice/{case}/data_small" in g.files:
        want = g[f"spice/{case}/data_small"]
        assert A.data_small.shape == want.shape
        assert np.array_equal(A.data_small, want, equal_nan=True)


@pytest.mark.parametrize("case", ["helio_serial", "helio_parallel", "helio_interval_cut_subfov", "helio_extend_pixel_size"])
def test_oracle_sweep_on_the_references_prepared_spice_image(case):
    """Sweep restatement (oracle) fed with the REFERENCE's prepared image and header: the reference's map."""
    from oracle import coreg_oracle as O
    g, m = load()
    c = m["spice"]["cases"][case]
    key = f"spice/{case}/data_small" if f"spice/{case}/data_small" in g.files else "spice/helio_serial/data_small"
    hs = {k: v for k, v in c["hdr_small"].items()}
    hl = dict(m["spice"]["hdr_large"])
    O.check_and_create_pcij_matrix(hl)
    ctor = c["ctor"]
    st = O.SweepState(hs, hl, g[key], g["spice/large"].astype(np.float64), ctor["lag_crval1"], ctor["lag_crval2"], None,
                      None, ctor["lag_crota"], unit_lag="arcsec", cdelt_semantics="reference")
    got = O.find_best_header_parameters(st, "helioprojective", parallelism=bool(ctor.get("parallelism")))
    want = g[f"spice/{case}/corr"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    tol = 0.0 if not ctor.get("parallelism") else 1e-10  # serial: float64 reference grid, bit-equal
    assert np.nanmax(np.abs(got - want)) <= max(tol, 1e-10), np.nanmax(np.abs(got - want))
    assert np.nanargmax(got) == np.nanargmax(want)


def test_reference_spice_carrington_entry_point_is_broken():
    """alignment_spice.py:146 names `self._carrington_transform`, which no class defines: the reference raises."""
    _, m = load()
    assert m["spice"]["cases"]["carrington_raises"]["raises"] == "AttributeError"


def test_raster_columns_take_the_frames_the_reference_took():
    """map_builder.py:95-99: mean time of each raster column (through the (x, y, t) WCS) -> nearest imager frame."""
    import datetime as dt
    from euispice_coreg_amd.utils import spice_header as S
    g, m = load()
    h4 = m["spice"]["hdr4d"]
    col_s, t_ref = S.column_times(h4)
    dates = [S.parse_date(h["DATE-AVG"]) for h in m["synras"]["imager_headers"]]
    chosen = [int(np.argmin([abs((t_ref + dt.timedelta(seconds=float(s)) - d).total_seconds()) for d in dates]))
              for s in col_s]
    assert chosen == m["synras"]["cases"]["process"]["frame_of_column"]
    assert m["synras"]["cases"]["threshold_raises"]["raises"] == "ValueError"
