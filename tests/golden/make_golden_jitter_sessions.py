#!/opt/conda/bin/python3.9
"""
Golden-vector generator: three more sessions of the REFERENCE's own
`euispice_coreg.jitter_correction.jitter_correction.jitter_correction_imagers` (`jitter_correction.py:14-174` sublists
and chain of corrected reference images, `:177-256` one alignment) on ONE seeded 6-image series, each with another
sublist layout -- `callers_golden` holds the plain case (sublists of 2, overlap 1):

    wide_overlap   sublist_length 3, overlap 2: images 3 and 4 are aligned on image 0 first, then image 4 again on the
                   corrected image 3 -- the second file overwrites the first
    one_sublist    sublist_length 10: every image against image 0, another Carrington grid, CROTA lags
    serial_minmax  sublist_length 2, overlap 1, `parallelism=False`, both pixel-value thresholds

    tests/golden/jitter_sessions_golden.npz    the six frames (float32, as the FITS files hold them)
    tests/golden/jitter_sessions_golden.json   headers as astropy read them back, the calls, the corrected header cards

Run (build container only; about a minute):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_jitter_sessions.py
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_callers as M  # noqa: E402  (loads the reference through _reference_loader)

import numpy as np  # noqa: E402
from astropy.io import fits  # noqa: E402

SEED = 99013


def main():
    tmp = tempfile.mkdtemp(prefix="golden_jitter_sessions_")
    series, jit = M.synthetic.make_series(n_frames=6, n=112, seed=SEED, jitter_sigma=3.0, n_blobs=160)
    ARR, META = {}, {"headers": [], "injected": M.jsonable(jit), "sessions": {}}
    paths = []
    for k, (img, h) in enumerate(series):
        ARR[f"frame{k}"] = np.asarray(img, dtype=np.float32)
        p = os.path.join(tmp, f"solo_L2_eui-hrieuv174-image_{k:03d}.fits")
        fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=img, header=M.to_header(h))]).writeto(p, overwrite=True)
        paths.append(p)
        with fits.open(p) as f:
            META["headers"].append(M.cards(f[-1].header))
    lag = [float(v) for v in np.arange(-10.0, 10.5, 2.0)]
    grid = dict(lonlims=[236.0, 256.0], latlims=[-4.0, 16.0], shape=[88, 88])
    sessions = {
        "wide_overlap": dict(grid, sublist_length=3, overlap=2, parallelism=True, cpu_count=4),
        "one_sublist": dict(lonlims=[238.0, 254.0], latlims=[-2.0, 14.0], shape=[72, 96], sublist_length=10, overlap=1,
                            lag_crota=[-0.1, 0.0, 0.1], parallelism=True, cpu_count=4),
        "serial_minmax": dict(grid, sublist_length=2, overlap=1, parallelism=False, small_fov_value_min=120.0,
                              small_fov_value_max=2500.0),
    }
    for name, kw in sessions.items():
        outdir = os.path.join(tmp, name)
        os.makedirs(outdir)
        call = dict(kw)
        arrays = {k: np.asarray(call.pop(k), dtype=np.float64) for k in ("lag_crota",) if k in call}
        M.jitter_correction_imagers(paths, outdir, lag_crval1=np.asarray(lag), lag_crval2=np.asarray(lag), **arrays, **call)
        outs = []
        for k, p in enumerate(paths):
            o = os.path.join(outdir, os.path.basename(p))
            with fits.open(o) as f:
                h = f[-1].header
                outs.append({"CRVAL1": float(h["CRVAL1"]), "CRVAL2": float(h["CRVAL2"]), "CROTA": float(h["CROTA"]),
                             "PC1_1": float(h["PC1_1"]), "PC1_2": float(h["PC1_2"]), "PC2_1": float(h["PC2_1"]),
                             "same_pixels": bool(np.array_equal(f[-1].data, series[k][0], equal_nan=True)),
                             "byte_copy_of_input": open(o, "rb").read() == open(p, "rb").read()})
            print(name, "frame", k, {kk: outs[-1][kk] for kk in ("CRVAL1", "CRVAL2", "CROTA")}, "injected",
                  jit[k].tolist(), flush=True)
        META["sessions"][name] = {"call": M.jsonable(dict(kw, lag_crval1=lag, lag_crval2=lag)), "outputs": outs}
    import astropy
    import scipy
    META["interpreter"] = {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                           "astropy": astropy.__version__, "seed": SEED}
    dst = os.path.join(HERE, "jitter_sessions_golden.npz")
    np.savez_compressed(dst, **ARR)
    with open(os.path.join(HERE, "jitter_sessions_golden.json"), "w") as f:
        json.dump(META, f, indent=1, sort_keys=True)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
