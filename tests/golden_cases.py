"""The reference-run fixtures (tests/golden/alignment_golden.{npz,json}, produced by tests/golden/make_golden_alignment.py
from the reference's own `Alignment`; `fixture="alignment_fuzz_golden"`: the seeded random family of
make_golden_alignment_fuzz.py, same layout) and the two ways the tests replay a case: through the oracle (CPU) and
through the product's drop-in `Alignment` on FITS files (GPU)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_cache = {}


DEFAULT = "alignment_golden"


def load(fixture=DEFAULT):
    if fixture not in _cache:
        g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
        with open(os.path.join(GOLDEN, fixture + ".json")) as f:
            _cache[fixture] = (g, json.load(f))
    return _cache[fixture]


# the reference returns a map there, but not one anybody should reproduce: dead workers leave their chunks at 0.0 (Q9)
ZEROS_LEFT_BY_DEAD_WORKERS = {"helio_cdelt2_parallel_zeros"}


def case_names(kind=None, fixture=DEFAULT):
    """kind: 'corr' (a map was returned), 'raises', 'results' (AlignmentResults surface), None = all."""
    _, m = load(fixture)
    out = []
    for name, c in sorted(m["cases"].items()):
        if name in ZEROS_LEFT_BY_DEAD_WORKERS:
            continue
        k = "raises" if "raises" in c else ("results" if "shift_arcsec" in c else "corr")
        if kind is None or k == kind:
            out.append(name)
    return out


def scene(name, fixture=DEFAULT):
    """(small float32, hdr_small, large float32, hdr_large) exactly as the reference read them from its FITS files."""
    g, m = load(fixture)
    sc = m["scenes"][name]
    return g[f"scene/{name}/small"], dict(sc["hdr_small"]), g[f"scene/{name}/large"], dict(sc["hdr_large"])


def expected(name, fixture=DEFAULT):
    g, m = load(fixture)
    return g[f"case/{name}/corr"], m["cases"][name]


FRAME = {"helioprojective": "helioprojective", "carrington": "carrington", "initial_carrington": "initial_carrington"}


def _deg(lims, unit):
    f = {"arcsec": 1.0 / 3600.0, "deg": 1.0}[unit]
    return [[v * f for v in lims[0]], [v * f for v in lims[1]]]


def oracle_replay(name, counts=None, hdr_small=None, fixture=DEFAULT):
    """The case through oracle/coreg_oracle.py, configured as the REFERENCE behaves (cdelt_semantics='reference',
    failed lag-points left as the reference leaves them).  Returns the 6-D map or raises what the oracle raises."""
    from oracle import coreg_oracle as O
    c = load(fixture)[1]["cases"][name]
    small, hs, large, hl = scene(c["scene"], fixture)
    if hdr_small is not None:
        hs = dict(hdr_small)
    ctor, call, ck = c["ctor"], c["call"], dict(c.get("call_kwargs") or {})
    if call == "carrington":
        if ck.get("method_carrington_reprojection", "fa") not in ("fa", "sunpy"):
            raise ValueError("method_carrington_reprojection must be either 'fa' or 'sunpy")  # alignment.py:189
    # alignment.py:191-198 / 301-314 (float64) / 371-384 (float32 for initial_carrington): exact either way
    small = small.astype(np.float64)
    large = large.astype(np.float64)
    shape = lonlims = latlims = None
    if call == "carrington":
        if ck.get("lonlims") is None and ck.get("latlims") is None and ck.get("size_deg_carrington") is not None:
            sd = ck["size_deg_carrington"]  # alignment.py:210-217
            lonlims = [hs["CRLN_OBS"] - 0.5 * sd[0], hs["CRLN_OBS"] + 0.5 * sd[0]]
            latlims = [hs["CRLT_OBS"] - 0.5 * sd[1], hs["CRLT_OBS"] + 0.5 * sd[1]]
            shape = [hs["NAXIS1"], hs["NAXIS2"]]
        elif ck.get("lonlims") is not None and ck.get("latlims") is not None and ck.get("shape") is not None:
            lonlims, latlims, shape = ck["lonlims"], ck["latlims"], ck["shape"]
        else:
            raise ValueError("either set lonlims as None, or not. no in between.")  # alignment.py:225
    O.check_and_create_pcij_matrix(hs, ctor.get("force_crota_0", False))
    O.check_and_create_pcij_matrix(hl, ctor.get("force_crota_0", False))
    st = O.SweepState(hs, hl, small, large, ctor.get("lag_crval1"), ctor.get("lag_crval2"), ctor.get("lag_cdelt1"),
                      ctor.get("lag_cdelt2"), ctor.get("lag_crota"), lag_solar_r=ctor.get("lag_solar_r"),
                      unit_lag=ctor.get("unit_lag", "arcsec"), order=ctor.get("reprojection_order", 2),
                      cdelt_semantics="reference")
    st.shape, st.lonlims, st.latlims = shape, lonlims, latlims
    # alignment.py:844-861
    O.set_threshold_minmax_to_nan(st.data_small, ctor.get("small_fov_value_min"), ctor.get("small_fov_value_max"))
    unit = ck.get("limits_unit", "arcsec")
    if ck.get("remove_fov_limits") is not None:
        r = _deg(ck["remove_fov_limits"], unit)
        O.set_remove_fov_limits_to_nan(st, r[0], r[1])
    if ck.get("fov_limits") is not None:
        r = _deg(ck["fov_limits"], unit)
        O.select_fov_in_small_data(st, r[0], r[1])
    return O.find_best_header_parameters(
        st, FRAME[call], method=ck.get("method", "correlation"), parallelism=bool(ctor.get("parallelism", False)),
        counts=counts, use_ang2pipi=(call != "initial_carrington"), reference_quirks=True)


def write_scene_fits(tmpdir, scene_name, fixture=DEFAULT):
    """The scene as FITS files written by THIS package's writer (no astropy on the GPU box): empty primary + float32
    image extension, the layout the generator used."""
    from euispice_coreg_amd.utils import fits_io
    small, hs, large, hl = scene(scene_name, fixture)
    ps, pl = os.path.join(str(tmpdir), scene_name + "_small.fits"), os.path.join(str(tmpdir), scene_name + "_large.fits")
    if not os.path.isfile(ps):
        fits_io.write_images(ps, [(None, {}), (small, hs)])
        fits_io.write_images(pl, [(None, {}), (large, hl)])
    return ps, pl


def product_replay(name, tmpdir, return_type="corr", fixture=DEFAULT, **extra):
    """The case through euispice_coreg_amd.hdrshift.Alignment (FITS in), configured to reproduce the reference
    (cdelt_semantics='reference')."""
    from euispice_coreg_amd.hdrshift import Alignment
    c = load(fixture)[1]["cases"][name]
    ps, pl = write_scene_fits(tmpdir, c["scene"], fixture)
    ctor = dict(c["ctor"])
    for k in ("lag_crval1", "lag_crval2", "lag_cdelt1", "lag_cdelt2", "lag_crota", "lag_solar_r"):
        if ctor.get(k) is not None:
            ctor[k] = np.asarray(ctor[k], dtype=np.float64)
    ck = dict(c.get("call_kwargs") or {})
    unit = ck.pop("limits_unit", None)  # plain numbers are read in the input lag unit by the product
    assert unit is None or unit == ctor.get("unit_lag", "arcsec")
    ctor.update(extra)
    A = Alignment(large_fov_known_pointing=pl, small_fov_to_correct=ps, cdelt_semantics="reference", **ctor)
    return A, getattr(A, "align_using_" + c["call"])(return_type=return_type, **ck)
