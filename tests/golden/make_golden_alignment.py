#!/opt/conda/bin/python3.9
"""
Golden-vector generator: runs the REFERENCE's own sweep driver end to end in the build container --
`euispice_coreg.hdrshift.alignment.Alignment` (`hdrshift/alignment.py:144-399` entry points, `:401-468` _shift_header,
`:509-578` _step / _step_no_shmm, `:613-797` _find_best_header_parameters, `:799-842`, `:844-887`, `:987-1029`,
`:1082-1127`), `hdrshift/c_correlate.py:39-72`, `hdrshift/AlignmentResults.py`, `utils/Util.py:106-215`
(write_corrected_fits / correct_pointing_header) -- on small seeded synthetic FITS files written by astropy, and stores
inputs + the reference's outputs as fixtures:

    tests/golden/alignment_golden.npz    images (float32, as the FITS files hold them) and correlation maps
    tests/golden/alignment_golden.json   headers as astropy read them back, the calls made, scalars, header cards, raises

Run (build container only; /root/reference must exist; about two minutes):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_alignment.py

The reference cannot travel to the GPU box; these files can.  Interpreter and load-time shims: see
`_reference_loader.py` (identity `numba.jit`, `multiprocess.shared_memory` alias, `from __future__ import annotations`
prepended at load, numpy-name shims, NEP-50 promotion).  Only reference modules are executed; nothing of them is copied.
The scenes come from this repo's own `euispice_coreg_amd/synthetic.py` (numpy only), loaded by path.
"""
import importlib.util
import json
import os
import sys
import tempfile
import traceback
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _reference_loader  # noqa: E402

_reference_loader.load_reference()
warnings.filterwarnings("ignore")

import numpy as np  # noqa: E402
import astropy.units as u  # noqa: E402
from astropy.io import fits  # noqa: E402

from euispice_coreg.hdrshift.alignment import Alignment  # noqa: E402

spec = importlib.util.spec_from_file_location("coreg_synthetic", os.path.join(ROOT, "euispice_coreg_amd", "synthetic.py"))
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)

STRUCTURAL = {"SIMPLE", "BITPIX", "NAXIS", "EXTEND", "XTENSION", "PCOUNT", "GCOUNT", "END", "COMMENT", "HISTORY", ""}
ARR = {}      # -> npz
META = {"scenes": {}, "cases": {}, "interpreter": {}}


def jsonable(v):
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.bool_,)):
        return bool(v)
    if isinstance(v, np.ndarray):
        return [jsonable(x) for x in v.tolist()]
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, u.Quantity):
        return jsonable(v.value)
    return v


def header_as_read(path, window):
    """Every non-structural card of the HDU as astropy reads it back (the values the reference works from)."""
    with fits.open(path) as hdul:
        h = hdul[window].header
        out = {}
        for k in h.keys():
            if k in STRUCTURAL or k.startswith("NAXIS"):
                continue
            out[k] = jsonable(h[k])
        out["NAXIS1"], out["NAXIS2"] = int(h["NAXIS1"]), int(h["NAXIS2"])
        return out


def write_pair(d, name, small, hs, large, hl):
    """FITS pair as the tests write it too: empty primary + one float32 image extension each."""
    ps, pl = os.path.join(d, name + "_small.fits"), os.path.join(d, name + "_large.fits")
    for p, img, h in ((ps, small, hs), (pl, large, hl)):
        hdr = fits.Header()
        for k, v in h.items():
            if k in ("NAXIS", "NAXIS1", "NAXIS2"):
                continue
            hdr[k] = v
        fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=np.asarray(img, dtype=np.float32), header=hdr)]).writeto(
            p, overwrite=True)
    ARR[f"scene/{name}/small"] = np.asarray(small, dtype=np.float32)
    ARR[f"scene/{name}/large"] = np.asarray(large, dtype=np.float32)
    META["scenes"][name] = {"hdr_small": header_as_read(ps, -1), "hdr_large": header_as_read(pl, -1)}
    return ps, pl


def quantity_limits(lims, unit):
    return None if lims is None else [u.Quantity(lims[0], unit), u.Quantity(lims[1], unit)]


def run_case(name, scene, paths, ctor, call, call_kwargs=None, note=""):
    """One reference call; `ctor` / `call_kwargs` hold plain numbers (units named separately) so that they are JSON."""
    ps, pl = paths
    call_kwargs = dict(call_kwargs or {})
    kw = dict(ctor)
    for k in ("lag_crval1", "lag_crval2", "lag_cdelt1", "lag_cdelt2", "lag_crota", "lag_solar_r"):
        if kw.get(k) is not None:
            kw[k] = np.asarray(kw[k], dtype=np.float64)
    ck = dict(call_kwargs)
    lim_unit = ck.pop("limits_unit", "arcsec")
    for k in ("fov_limits", "remove_fov_limits"):
        if k in ck:
            ck[k] = quantity_limits(ck[k], lim_unit)
    entry = {"scene": scene, "ctor": jsonable(ctor), "call": call, "call_kwargs": jsonable(call_kwargs), "note": note}
    try:
        A = Alignment(large_fov_known_pointing=pl, small_fov_to_correct=ps, **kw)
        corr = getattr(A, "align_using_" + call)(return_type="corr", **ck)
        ARR[f"case/{name}/corr"] = np.asarray(corr, dtype=np.float64)
        entry["shape"] = list(corr.shape)
        entry["nan_count"] = int(np.isnan(corr).sum())
        if np.isfinite(corr).any():
            entry["argmax"] = [int(v) for v in np.unravel_index(np.nanargmax(corr), corr.shape)]
        # what the reference's state ended as (lags converted to header units, alignment.py:819-837)
        entry["state"] = {"unit_lag": A.unit_lag, "lag_crval1": jsonable(np.asarray(A.lag_crval1)),
                          "lag_crval2": jsonable(np.asarray(A.lag_crval2)),
                          "lag_solar_r": jsonable(np.asarray(A.lag_solar_r)),
                          "crota_ref": jsonable(A.crota_ref), "data_small_shape": None}
        print(f"{name:32s} corr {tuple(corr.shape)} nan {entry['nan_count']:4d} max {np.nanmax(corr) if np.isfinite(corr).any() else float('nan'):.6f}",
              flush=True)
    except Exception as e:  # the raise IS the fixture
        entry["raises"] = type(e).__name__
        entry["message"] = str(e)[:200]
        print(f"{name:32s} raises {type(e).__name__}: {str(e)[:80]}", flush=True)
    META["cases"][name] = entry
    return entry


def results_case(name, scene, paths, ctor, call, call_kwargs, tmp):
    """AlignmentResults surface (AlignmentResults.py:24-101, 218-341, 150-215) + Util.write_corrected_fits."""
    ps, pl = paths
    kw = {k: (np.asarray(v, dtype=np.float64) if k.startswith("lag_") and v is not None else v) for k, v in ctor.items()}
    A = Alignment(large_fov_known_pointing=pl, small_fov_to_correct=ps, **kw)
    res = getattr(A, "align_using_" + call)(**call_kwargs)
    ARR[f"case/{name}/corr"] = np.asarray(res.corr, dtype=np.float64)
    hdr = res.return_corrected_header(window=-1)
    out = os.path.join(tmp, name + "_corrected.fits")
    res.write_corrected_fits(window_list_to_apply_shift=[-1], path_to_l3_output=out)
    with fits.open(out) as hdul:
        written = {k: jsonable(hdul[-1].header[k]) for k in ("CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "CROTA", "PC1_1",
                                                               "PC1_2", "PC2_1", "PC2_2")}
        same_pixels = bool(np.array_equal(hdul[-1].data, fits.getdata(ps, -1), equal_nan=True))
        n_hdu = len(hdul)
    META["cases"][name] = {
        "scene": scene, "ctor": jsonable(ctor), "call": call, "call_kwargs": jsonable(call_kwargs),
        "shape": list(res.corr.shape), "max_index": [int(v) for v in res.max_index],
        "shift_pixels": jsonable(list(res.shift_pixels)), "shift_arcsec": jsonable(list(res.shift_arcsec)),
        "unit_lag": res.unit_lag,
        "parameters_alignment_arcsec": {k: jsonable(v) for k, v in res.parameters_alignment_arcsec.items()},
        "corrected_header": {k: jsonable(hdr[k]) for k in ("CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "CROTA", "PC1_1",
                                                              "PC1_2", "PC2_1", "PC2_2")},
        "written_header": written, "written_same_pixels": same_pixels, "written_n_hdu": n_hdu,
    }
    print(f"{name:32s} shift_arcsec {tuple(float(v) for v in res.shift_arcsec)}", flush=True)


def main():
    tmp = tempfile.mkdtemp(prefix="golden_alignment_")
    lon, lat, shape = [228.0, 262.0], [-12.0, 22.0], [72, 64]
    carr = {"lonlims": lon, "latlims": lat, "shape": shape}

    # ---- scene A: tests.helpers.scene(96, 160): HRIEUV-like / FSI-like pair, pointing error (17, -9) arcsec, 0.3 deg
    small, hs, large, hl, _ = synthetic.make_scene(small_n=96, large_n=160, seed=5, n_blobs=120)
    A_ = write_pair(tmp, "A", small, hs, large, hl)
    l1, l2 = [5.0, 9.0, 13.0, 17.0, 21.0, 25.0, 29.0], [-19.0, -14.0, -9.0, -4.0, 1.0, 6.0]
    base = dict(lag_crval1=l1, lag_crval2=l2, lag_cdelt1=[0.0], lag_cdelt2=[0.0], lag_crota=[0.0, 0.3])
    for par in (False, True):
        tag = "parallel" if par else "serial"
        run_case(f"helio_{tag}", "A", A_, dict(base, parallelism=par, counts_cpu_max=4), "helioprojective",
                 note="quirk Q1: serial = full reference grid (float64), parallel = sub-map on the small grid (float32)")
        run_case(f"carr_{tag}", "A", A_, dict(base, parallelism=par, counts_cpu_max=4), "carrington", carr)
    # zero lag on both axes (identical headers: wcslib's rounding noise decides the border pixels)
    run_case("helio_zero_lag", "A", A_, dict(lag_crval1=[-4.0, 0.0, 4.0], lag_crval2=[-4.0, 0.0, 4.0], lag_cdelt1=None,
                                             lag_cdelt2=None, lag_crota=None, parallelism=True, counts_cpu_max=3),
             "helioprojective")
    # quirk Q2 / Q3: a CDELT1 lag never reaches the header but rebuilds PC; a CDELT2 lag raises
    cd = dict(lag_crval1=[13.0, 17.0, 21.0], lag_crval2=[-13.0, -9.0, -5.0], lag_cdelt1=[-0.05, 0.0, 0.05],
              lag_cdelt2=[0.0], lag_crota=[0.3])
    run_case("helio_cdelt1_serial", "A", A_, dict(cd, parallelism=False), "helioprojective")
    run_case("helio_cdelt1_parallel", "A", A_, dict(cd, parallelism=True, counts_cpu_max=4), "helioprojective")
    run_case("carr_cdelt1_serial", "A", A_, dict(cd, parallelism=False), "carrington", carr)
    run_case("helio_cdelt2_raises", "A", A_, dict(cd, lag_cdelt1=[0.0], lag_cdelt2=[0.0, 0.05], parallelism=False),
             "helioprojective")
    # the same lag set through the parallel branch: the worker that meets the first d_cdelt2 != 0 dies and its whole
    # np.array_split chunk -- valid lag-points behind it included -- keeps the initial 0.0 (quirk Q9)
    run_case("helio_cdelt2_parallel_zeros", "A", A_, dict(cd, lag_cdelt1=[0.0], lag_cdelt2=[0.0, 0.05], parallelism=True,
                                                          counts_cpu_max=3), "helioprojective",
             note="quirk Q9: failed workers leave zeros")
    # thresholds + order 1
    th = dict(lag_crval1=l1[1:6], lag_crval2=l2[1:5], lag_cdelt1=None, lag_cdelt2=None, lag_crota=[0.3],
              small_fov_value_min=150.0, small_fov_value_max=900.0, reprojection_order=1)
    run_case("helio_thresholds_order1", "A", A_, dict(th, parallelism=True, counts_cpu_max=4), "helioprojective")
    run_case("carr_thresholds_order1", "A", A_, dict(th, parallelism=False), "carrington", carr)
    run_case("helio_all_nan_raises", "A", A_, dict(th, small_fov_value_min=1e9, parallelism=True, counts_cpu_max=2),
             "helioprojective")
    # order 3 (serial: target = the reference image's own grid)
    run_case("helio_order3_serial", "A", A_, dict(th, reprojection_order=3, small_fov_value_min=None, parallelism=False),
             "helioprojective")
    # lags given in degrees against a header in arcsec (alignment.py:819-837)
    run_case("helio_unit_lag_deg", "A", A_, dict(lag_crval1=[v / 3600.0 for v in l1[1:6]],
                                                 lag_crval2=[v / 3600.0 for v in l2[1:5]], lag_cdelt1=None,
                                                 lag_cdelt2=None, lag_crota=[0.3], unit_lag="deg", parallelism=True,
                                                 counts_cpu_max=4), "helioprojective")
    # method 'residus' (quirk Q8: no NaN mask)
    run_case("helio_residus", "A", A_, dict(th, reprojection_order=2, parallelism=True, counts_cpu_max=4),
             "helioprojective", {"method": "residus"})
    run_case("carr_residus_serial", "A", A_, dict(th, reprojection_order=2, parallelism=False), "carrington",
             dict(carr, method="residus"))
    # fov_limits / remove_fov_limits (alignment.py:844-874, 1082-1127), limits in arcsec
    fl = dict(lag_crval1=l1[1:6], lag_crval2=l2[1:5], lag_cdelt1=None, lag_cdelt2=None, lag_crota=None, parallelism=True,
              counts_cpu_max=4)
    run_case("helio_fov_limits", "A", A_, fl, "helioprojective",
             {"fov_limits": [[-700.0, -100.0], [100.0, 700.0]], "limits_unit": "arcsec"})
    run_case("helio_remove_fov_limits", "A", A_, fl, "helioprojective",
             {"remove_fov_limits": [[-450.0, -300.0], [350.0, 520.0]], "limits_unit": "arcsec"})
    run_case("helio_fov_and_remove", "A", A_, fl, "helioprojective",
             {"fov_limits": [[-700.0, -100.0], [100.0, 700.0]], "remove_fov_limits": [[-450.0, -300.0], [350.0, 520.0]],
              "limits_unit": "arcsec"})
    # two lag_solar_r values, serial: the second pass reprojects the ALREADY reprojected reference (alignment.py:761-764)
    run_case("carr_two_solar_r_serial", "A", A_, dict(lag_crval1=l1[2:5], lag_crval2=l2[1:4], lag_cdelt1=None,
                                                      lag_cdelt2=None, lag_crota=None, lag_solar_r=[1.004, 1.02],
                                                      parallelism=False), "carrington", carr,
             note="quirk Q10: slice 1 of the last axis correlates against a twice-reprojected reference")
    run_case("carr_size_deg", "A", A_, dict(lag_crval1=l1[2:5], lag_crval2=l2[1:4], lag_cdelt1=None, lag_cdelt2=None,
                                            lag_crota=None, parallelism=False), "carrington",
             {"size_deg_carrington": [30.0, 30.0]}, note="alignment.py:210-217: grid from CRLN_OBS / CRLT_OBS")
    run_case("carr_bad_grid_raises", "A", A_, dict(lag_crval1=[0.0], lag_crval2=[0.0], lag_cdelt1=None, lag_cdelt2=None,
                                                   lag_crota=None), "carrington", {"lonlims": lon})
    run_case("carr_sunpy_method_raises", "A", A_, dict(lag_crval1=[0.0], lag_crval2=[0.0], lag_cdelt1=None,
                                                       lag_cdelt2=None, lag_crota=None), "carrington",
             dict(carr, method_carrington_reprojection="other"))

    # ---- scene B: PCi_j NOT consistent with CROTA (quirk Q3: the d_crota == 0 slice keeps the header's own PC)
    hb = dict(hs)
    rho, lam = np.deg2rad(3.04), hs["CDELT2"] / hs["CDELT1"]
    hb.update(PC1_1=float(np.cos(rho)), PC2_2=float(np.cos(rho)), PC1_2=float(-lam * np.sin(rho)),
              PC2_1=float(np.sin(rho) / lam))
    B_ = write_pair(tmp, "B", small, hb, large, hl)
    q3 = dict(lag_crval1=[13.0, 17.0, 21.0], lag_crval2=[-13.0, -9.0, -5.0], lag_cdelt1=None, lag_cdelt2=None,
              lag_crota=[-0.2, 0.0, 0.2, 0.3])
    run_case("helio_pc_inconsistent_serial", "B", B_, dict(q3, parallelism=False), "helioprojective")
    run_case("helio_pc_inconsistent_parallel", "B", B_, dict(q3, parallelism=True, counts_cpu_max=4), "helioprojective")
    run_case("carr_pc_inconsistent", "B", B_, dict(q3, parallelism=False), "carrington", carr,
             note="quirk Q4: the Carrington path ignores PCi_j")

    # ---- scene C: CROTA2 only, no CROTA, no PCi_j (alignment.py:580-611, 445-449)
    hc = {k: v for k, v in hs.items() if k not in ("PC1_1", "PC1_2", "PC2_1", "PC2_2", "CROTA")}
    hc["CROTA2"] = 3.0
    C_ = write_pair(tmp, "C", small, hc, large, hl)
    run_case("helio_crota2_only", "C", C_, dict(q3, parallelism=True, counts_cpu_max=4), "helioprojective")
    run_case("carr_crota2_only", "C", C_, dict(q3, parallelism=False), "carrington", carr)

    # ---- scene D: neither CROTA, CROTA2 nor PCi_j: ValueError unless force_crota_0 (alignment.py:587-593)
    hd = {k: v for k, v in hs.items() if k not in ("PC1_1", "PC1_2", "PC2_1", "PC2_2", "CROTA")}
    D_ = write_pair(tmp, "D", small, hd, large, hl)
    f0 = dict(lag_crval1=l1[1:6], lag_crval2=l2[1:5], lag_cdelt1=None, lag_cdelt2=None, lag_crota=[0.0, 3.3])
    run_case("helio_no_rotation_raises", "D", D_, dict(f0, parallelism=False), "helioprojective")
    run_case("helio_force_crota_0", "D", D_, dict(f0, parallelism=True, counts_cpu_max=4, force_crota_0=True),
             "helioprojective")
    run_case("carr_force_crota_0", "D", D_, dict(f0, parallelism=False, force_crota_0=True), "carrington", carr)

    # ---- scene E: header of the image to align in DEGREES (SPICE-like), lags in arcsec
    sm_e, hs_e, lg_e, hl_e, _ = synthetic.make_scene(small_n=96, large_n=160, seed=9, n_blobs=120, small_unit="deg",
                                                     small_shape=(112, 40), small_cdelt=(4.0, 1.098))
    E_ = write_pair(tmp, "E", sm_e, hs_e, lg_e, hl_e)
    de = dict(lag_crval1=[9.0, 13.0, 17.0, 21.0, 25.0], lag_crval2=[-17.0, -13.0, -9.0, -5.0, -1.0], lag_cdelt1=None,
              lag_cdelt2=None, lag_crota=[0.0, 0.3])
    run_case("helio_header_deg_parallel", "E", E_, dict(de, parallelism=True, counts_cpu_max=4), "helioprojective")
    run_case("helio_header_deg_serial", "E", E_, dict(de, parallelism=False), "helioprojective")

    # ---- scene F: two Carrington maps (align_using_initial_carrington, alignment.py:344-399), degrees
    sm_f, hs_f, lg_f, hl_f, _ = synthetic.make_car_scene(small_shape=(60, 80), large_shape=(100, 130))
    F_ = write_pair(tmp, "F", sm_f, hs_f, lg_f, hl_f)
    lf = dict(lag_crval1=[0.0, 0.006, 0.012, 0.018, 0.024], lag_crval2=[-0.022, -0.011, 0.0, 0.011],
              lag_cdelt1=None, lag_cdelt2=None, lag_crota=None, unit_lag="deg")
    run_case("initial_carrington_parallel", "F", F_, dict(lf, parallelism=True, counts_cpu_max=4), "initial_carrington")
    run_case("initial_carrington_serial", "F", F_, dict(lf, parallelism=False), "initial_carrington")

    # ---- AlignmentResults + corrected FITS
    rl1, rl2 = [float(v) for v in np.arange(9.0, 26.0, 2.0)], [float(v) for v in np.arange(-17.0, 0.0, 2.0)]
    results_case("results_helio", "A", A_, dict(lag_crval1=rl1, lag_crval2=rl2, lag_cdelt1=None, lag_cdelt2=None,
                                                lag_crota=[0.0, 0.3], parallelism=True, counts_cpu_max=4),
                 "helioprojective", {}, tmp)
    results_case("results_carr", "A", A_, dict(lag_crval1=rl1, lag_crval2=rl2, lag_cdelt1=None, lag_cdelt2=None,
                                               lag_crota=[0.0, 0.3], parallelism=True, counts_cpu_max=4),
                 "carrington", carr, tmp)
    results_case("results_helio_header_deg", "E", E_, dict(de, parallelism=True, counts_cpu_max=4), "helioprojective", {},
                 tmp)

    import astropy
    import scipy
    META["interpreter"] = {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                           "astropy": astropy.__version__}
    np.savez_compressed(os.path.join(HERE, "alignment_golden.npz"), **ARR)
    with open(os.path.join(HERE, "alignment_golden.json"), "w") as f:
        json.dump(META, f, indent=1, sort_keys=True)
    print("wrote alignment_golden.npz", os.path.getsize(os.path.join(HERE, "alignment_golden.npz")), "bytes,",
          len(META["cases"]), "cases")


if __name__ == "__main__":
    try:
        main()
    except Exception:
        traceback.print_exc()
        sys.exit(1)
