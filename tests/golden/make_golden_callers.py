#!/opt/conda/bin/python3.9
"""
Golden-vector generator for the callers either side of the sweep: runs the REFERENCE's own

  * `hdrshift.alignment_spice.AlignmentSpice` (`alignment_spice.py:66-120` entry point, `:189-221`, `:223-248`,
    `:250-323` _prepare_spice_from_l2) on a synthetic SPICE L2 window,
  * `synras.map_builder.SPICEComposedMapBuilder.process` (`map_builder.py:57-79`, `:87-214`, `:249-294`),
  * `jitter_correction.jitter_correction.jitter_correction_imagers` (`jitter_correction.py:14-174`, `:177-256`)

in the build container, on FITS files written by astropy, and stores inputs + outputs:

    tests/golden/callers_golden.npz    input images (float32 / the float64 factors the cube is built from), rasters, maps
    tests/golden/callers_golden.json   headers as astropy read them back, calls, header cards, scalars, raises

Run (build container only; about a minute):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_callers.py

Interpreter and load-time shims: `_reference_loader.py`.  Only reference modules are executed; nothing is copied.
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

from euispice_coreg.hdrshift.alignment_spice import AlignmentSpice  # noqa: E402
from euispice_coreg.synras.map_builder import SPICEComposedMapBuilder  # noqa: E402
from euispice_coreg.jitter_correction.jitter_correction import jitter_correction_imagers  # noqa: E402

spec = importlib.util.spec_from_file_location("coreg_synthetic", os.path.join(ROOT, "euispice_coreg_amd", "synthetic.py"))
synthetic = importlib.util.module_from_spec(spec)
spec.loader.exec_module(synthetic)

STRUCTURAL = {"SIMPLE", "BITPIX", "EXTEND", "XTENSION", "PCOUNT", "GCOUNT", "END", "COMMENT", "HISTORY", ""}
ARR, META = {}, {"spice": {}, "synras": {}, "jitter": {}}


def jsonable(v):
    if isinstance(v, np.floating):
        return float(v)
    if isinstance(v, np.integer):
        return int(v)
    if isinstance(v, np.bool_):
        return bool(v)
    if isinstance(v, np.ndarray):
        return [jsonable(x) for x in v.tolist()]
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, u.Quantity):
        return jsonable(v.value)
    return v


def cards(h):
    return {k: jsonable(h[k]) for k in h.keys() if k not in STRUCTURAL}


def to_header(d):
    h = fits.Header()
    for k, v in d.items():
        if k == "NAXIS" or k.startswith("NAXIS"):
            continue
        h[k] = v
    return h


def main():
    tmp = tempfile.mkdtemp(prefix="golden_callers_")
    # ---------------------------------------------------------------------------------------------------- SPICE L2
    cube, h4, large, hl, truth = synthetic.make_spice_l2(nw=8)
    # the cube is rebuilt by the tests from these two factors (IEEE multiplication + float32 rounding: deterministic)
    ARR["spice/image"] = np.asarray(truth["image"], dtype=np.float64)
    ARR["spice/profile"] = np.asarray(truth["profile"], dtype=np.float64)
    rebuilt = (ARR["spice/image"][None, :, :] * ARR["spice/profile"][:, None, None])[None].astype(np.float32)
    assert np.array_equal(rebuilt, cube)
    large32 = np.asarray(large, dtype=np.float32)
    ARR["spice/large"] = large32
    p_spice = os.path.join(tmp, "solo_L2_spice-n-ras_20220317T094045_V01.fits")
    fits.HDUList([fits.PrimaryHDU(data=cube, header=to_header(h4))]).writeto(p_spice, overwrite=True)
    p_large = os.path.join(tmp, "solo_L2_eui-fsi174-image_ref.fits")
    fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=large32, header=to_header(hl))]).writeto(p_large, overwrite=True)
    with fits.open(p_spice) as f:
        META["spice"]["hdr4d"] = cards(f[0].header)
    with fits.open(p_large) as f:
        META["spice"]["hdr_large"] = cards(f[-1].header)
    l1 = [float(v) for v in np.arange(-35.0, -10.0, 4.0)]
    l2 = [float(v) for v in np.arange(24.0, 49.0, 4.0)]
    META["spice"]["cases"] = {}

    def spice_case(name, ctor, call, call_kwargs, quantities=None):
        kw = dict(ctor)
        for k in ("lag_crval1", "lag_crval2", "lag_crota"):
            if kw.get(k) is not None:
                kw[k] = np.asarray(kw[k], dtype=np.float64)
        if quantities:  # plain numbers in the JSON, Quantities in the call
            if "wavelength_interval_to_sum" in quantities:
                kw["wavelength_interval_to_sum"] = [v * u.angstrom for v in quantities["wavelength_interval_to_sum"]]
            if "sub_fov_window" in quantities:
                kw["sub_fov_window"] = [v * u.arcsec for v in quantities["sub_fov_window"]]
        entry = {"ctor": jsonable(ctor), "call": call, "call_kwargs": jsonable(call_kwargs),
                 "quantities": jsonable(quantities or {})}
        try:
            A = AlignmentSpice(large_fov_known_pointing=p_large, small_fov_to_correct=p_spice, small_fov_window=0,
                               large_fov_window=-1, **kw)
            corr = getattr(A, "align_using_" + call)(return_type="corr", **call_kwargs)
            ARR[f"spice/{name}/corr"] = np.asarray(corr, dtype=np.float64)
            entry["shape"] = list(corr.shape)
            entry["hdr_small"] = cards(A.hdr_small)
            if not kw.get("parallelism"):  # the serial branch keeps the prepared image (alignment.py:660-665 deletes it)
                ARR[f"spice/{name}/data_small"] = np.asarray(A.data_small, dtype=np.float64)
            print(f"{name:28s} corr {tuple(corr.shape)} max {np.nanmax(corr):.6f} argmax "
                  f"{np.unravel_index(np.nanargmax(corr), corr.shape)[:2]}", flush=True)
        except Exception as e:
            entry["raises"] = type(e).__name__
            entry["message"] = str(e)[:200]
            print(f"{name:28s} raises {type(e).__name__}: {str(e)[:90]}", flush=True)
        META["spice"]["cases"][name] = entry

    base = dict(lag_crval1=l1, lag_crval2=l2, lag_crota=[0.0])
    spice_case("helio_serial", dict(base, parallelism=False), "helioprojective", {})
    spice_case("helio_parallel", dict(base, parallelism=True, counts_cpu_max=4), "helioprojective", {})
    wave = np.asarray(ARR["spice/profile"].shape)  # (placeholder so the name is used)
    del wave
    # wavelength interval: planes 2..5 of 8 (CRVAL3 97.7031 nm, CDELT3 0.00973 nm, CRPIX3 4.5)
    w = (97.7031 + 0.00973 * (np.arange(8) + 1 - 4.5)) * 10.0  # angstrom
    spice_case("helio_interval_cut_subfov", dict(base, parallelism=False), "helioprojective",
               {"cut_from_center": 30}, {"wavelength_interval_to_sum": [float(w[2] - 1e-4), float(w[5] + 1e-4)],
                                         "sub_fov_window": [-420.0, -230.0, 340.0, 500.0]})
    spice_case("helio_extend_pixel_size", dict(base, parallelism=False), "helioprojective", {"extend_pixel_size": True})
    spice_case("carrington_raises", dict(base, parallelism=False), "carrington",
               {"lonlims": [228.0, 262.0], "latlims": [-12.0, 22.0], "shape": [64, 64]})

    # ---------------------------------------------------------------------------------------------------- synras
    frames = synthetic.make_imager_sequence(large32.astype(np.float64), hl)
    paths = []
    for k, (img, h) in enumerate(frames):
        p = os.path.join(tmp, f"solo_L2_eui-fsi174-image_{k:02d}.fits")
        fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=img, header=to_header(h))]).writeto(p, overwrite=True)
        paths.append(p)
    META["synras"]["imager_headers"] = []
    for p in paths:
        with fits.open(p) as f:
            META["synras"]["imager_headers"].append(cards(f[-1].header))
    META["synras"]["cases"] = {}
    for name, kw in (("process", {}), ("keep_pixel_size", {"keep_original_imager_pixel_size": True})):
        C = SPICEComposedMapBuilder(path_to_spectro=p_spice, list_imager_paths=paths, threshold_time=u.Quantity(200, "s"),
                                    window_imager=-1, window_spectro=0)
        out = C.process(folder_path_output=tmp, basename_output=f"synras_{name}.fits", print_filename=False,
                        return_synras_name=True, **kw)
        with fits.open(out) as f:
            ARR[f"synras/{name}/raster"] = np.asarray(f[0].data, dtype=np.float64)
            META["synras"]["cases"][name] = {
                "kwargs": kw, "header": cards(f[0].header), "shape": list(f[0].data.shape),
                "frame_of_column": [int(np.argmin([abs((d - t).to("s").value) for t in C.dates]))
                                    for d in C.dates_selected]}
        print(f"synras {name:18s} raster {ARR[f'synras/{name}/raster'].shape} frames "
              f"{sorted(set(META['synras']['cases'][name]['frame_of_column']))}", flush=True)
    try:
        SPICEComposedMapBuilder(p_spice, paths[:2], threshold_time=u.Quantity(100, "s"), window_spectro=0).process(
            folder_path_output=tmp, basename_output="x.fits", print_filename=False)
        META["synras"]["cases"]["threshold_raises"] = {"raises": None}
    except Exception as e:
        META["synras"]["cases"]["threshold_raises"] = {"raises": type(e).__name__}
    # the synthetic raster as reference image of AlignmentSpice (README example of the reference)
    A = AlignmentSpice(os.path.join(tmp, "synras_process.fits"), p_spice, lag_crval1=np.arange(-30.0, -14.0, 2.5),
                       lag_crval2=np.arange(28.0, 45.0, 2.5), large_fov_window=0, small_fov_window=0, parallelism=False)
    corr = A.align_using_helioprojective(return_type="corr")
    ARR["synras/align_on_raster/corr"] = np.asarray(corr, dtype=np.float64)
    META["synras"]["cases"]["align_on_raster"] = {"lag_crval1": jsonable(np.arange(-30.0, -14.0, 2.5)),
                                                  "lag_crval2": jsonable(np.arange(28.0, 45.0, 2.5))}
    print("synras align_on_raster max", float(np.nanmax(corr)), flush=True)

    # ---------------------------------------------------------------------------------------------------- jitter
    series, jit = synthetic.make_series(n_frames=4, n=128, seed=7, jitter_sigma=4.0)
    jpaths = []
    for k, (img, h) in enumerate(series):
        ARR[f"jitter/frame{k}"] = np.asarray(img, dtype=np.float32)
        p = os.path.join(tmp, f"solo_L2_eui-hrieuv174-image_{k:03d}.fits")
        fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=img, header=to_header(h))]).writeto(p, overwrite=True)
        jpaths.append(p)
    META["jitter"]["headers"] = []
    for p in jpaths:
        with fits.open(p) as f:
            META["jitter"]["headers"].append(cards(f[-1].header))
    META["jitter"]["injected"] = jsonable(jit)
    lag = [float(v) for v in np.arange(-12.0, 12.5, 2.0)]
    kw = dict(lonlims=[236.0, 256.0], latlims=[-4.0, 16.0], shape=[96, 96], sublist_length=2, overlap=1,
              small_fov_value_max=2800.0)
    outdir = os.path.join(tmp, "jitter_out")
    os.makedirs(outdir)
    jitter_correction_imagers(jpaths, outdir, lag_crval1=np.asarray(lag), lag_crval2=np.asarray(lag), parallelism=True,
                              cpu_count=4, **kw)
    META["jitter"]["call"] = dict(kw, lag_crval1=lag, lag_crval2=lag)
    META["jitter"]["outputs"] = []
    for k, p in enumerate(jpaths):
        o = os.path.join(outdir, os.path.basename(p))
        with fits.open(o) as f:
            h = f[-1].header
            META["jitter"]["outputs"].append({
                "CRVAL1": float(h["CRVAL1"]), "CRVAL2": float(h["CRVAL2"]), "CROTA": float(h["CROTA"]),
                "PC1_1": float(h["PC1_1"]), "PC1_2": float(h["PC1_2"]),
                "same_pixels": bool(np.array_equal(f[-1].data, series[k][0], equal_nan=True)),
                "byte_copy_of_input": open(o, "rb").read() == open(p, "rb").read()})
        print("jitter frame", k, META["jitter"]["outputs"][-1], "injected", jit[k].tolist(), flush=True)

    np.savez_compressed(os.path.join(HERE, "callers_golden.npz"), **ARR)
    with open(os.path.join(HERE, "callers_golden.json"), "w") as f:
        json.dump(META, f, indent=1, sort_keys=True)
    print("wrote callers_golden.npz", os.path.getsize(os.path.join(HERE, "callers_golden.npz")), "bytes")


if __name__ == "__main__":
    try:
        main()
    except Exception:
        traceback.print_exc()
        sys.exit(1)
