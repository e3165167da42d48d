#!/opt/conda/bin/python3.9
"""
Golden-vector generator: the REFERENCE's own `AlignmentResults` (`hdrshift/AlignmentResults.py:24-101` argmax and lag
bookkeeping, `:218-341` sub-lag Gaussian fit, `:196-215` return_corrected_header, `:149-180` write_corrected_fits ->
`utils/Util.py:106-215`) on 5-D correlation maps WITH CDELT AND CROTA AXES.  The reference's sweep cannot produce such a
map (a CDELT2 lag kills its worker, quirk Q2), so the reference-run `results_*` cases of `alignment_golden` all have
d_cdelt = 0; but `AlignmentResults` takes any map, and what it does with the CDELT / CROTA part of the argmax -- hand it to
`correct_pointing_header` -- is exactly the intended CDELT semantics of the sweep.  Six seeded maps: a Gaussian peak over
(CRVAL1, CRVAL2) between lag values x a profile over (CDELT1, CDELT2, CROTA) with its maximum at chosen indices + 1e-4 of
noise; lags in arcsec and in degrees; headers in arcsec and in degrees (scenes A and E of `alignment_golden`, read from
that fixture so that the pixels and cards are the same).

    tests/golden/results_5d_golden.npz    the maps
    tests/golden/results_5d_golden.json   lag axes, the reference's max_index / shift_pixels / shift_arcsec /
                                          parameters_alignment_arcsec, the corrected header it returns and the one it writes

Run (build container only; seconds):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_results_5d.py
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_alignment as M  # noqa: E402  (loads the reference through _reference_loader)

import numpy as np  # noqa: E402
from astropy.io import fits  # noqa: E402
from euispice_coreg.hdrshift.AlignmentResults import AlignmentResults  # noqa: E402

CARDS = ("CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "CROTA", "PC1_1", "PC1_2", "PC2_1", "PC2_2")


def make_map(rng, axes, peak, widths):
    """corr[i1, i2, i3, i4, i5, 0]: Gaussian in the two CRVAL lags about `peak[:2]` (lag VALUES, between samples), times a
    profile over the other three axes that peaks at the INDICES peak[2:]."""
    l1, l2 = np.asarray(axes[0]), np.asarray(axes[1])
    g = np.exp(-0.5 * (((l1[:, None] - peak[0]) / widths[0]) ** 2 + ((l2[None, :] - peak[1]) / widths[1]) ** 2))
    prof = np.ones(tuple(len(a) for a in axes[2:]))
    for ax, (n, k) in enumerate(zip(prof.shape, peak[2:])):
        w = 1.0 - 0.08 * np.abs(np.arange(n) - k) ** 1.5
        prof = prof * w.reshape([-1 if a == ax else 1 for a in range(3)])
    corr = 0.15 + 0.8 * g[:, :, None, None, None] * prof[None, None]
    corr = corr + 1e-4 * rng.standard_normal(corr.shape)
    return corr[..., None]


def main():
    tmp = tempfile.mkdtemp(prefix="golden_results_5d_")
    g0 = np.load(os.path.join(HERE, "alignment_golden.npz"))
    m0 = json.load(open(os.path.join(HERE, "alignment_golden.json")))
    ARR, META = {}, {"cases": {}, "scenes": {}}
    paths = {}
    for sc in ("A", "E"):
        hs = m0["scenes"][sc]["hdr_small"]
        p = os.path.join(tmp, sc + "_small.fits")
        hdr = fits.Header()
        for k, v in hs.items():
            if not k.startswith("NAXIS"):
                hdr[k] = v
        fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=g0[f"scene/{sc}/small"], header=hdr)]).writeto(p, overwrite=True)
        paths[sc] = p
        META["scenes"][sc] = {"hdr_small": M.header_as_read(p, -1)}
    cases = [
        # name, scene, unit_lag, axes (crval1, crval2, cdelt1, cdelt2, crota), peak (values, values, index, index, index)
        ("arcsec_full", "A", "arcsec", (np.arange(9.0, 26.0, 2.0), np.arange(-17.0, 0.0, 2.0), [-0.05, 0.0, 0.05, 0.1],
                                        [-0.04, 0.0, 0.04], [-0.2, 0.0, 0.3]), (17.6, -8.7, 2, 0, 2), (3.1, 2.6)),
        ("arcsec_cdelt1_only", "A", "arcsec", (np.arange(9.0, 26.0, 2.0), np.arange(-17.0, 0.0, 2.0), [-0.05, 0.0, 0.05],
                                               [0.0], [0.0]), (15.2, -9.9, 0, 0, 0), (2.7, 3.3)),
        ("arcsec_cdelt2_crota", "A", "arcsec", (np.arange(5.0, 30.0, 3.0), np.arange(-20.0, 3.0, 3.0), [0.0],
                                                [-0.02, 0.0, 0.02, 0.04], [0.0, 0.25, 0.5]), (18.9, -7.1, 0, 3, 1), (4.0, 4.4)),
        ("deg_header_arcsec_lags", "E", "arcsec", (np.arange(9.0, 26.0, 2.0), np.arange(-17.0, 0.0, 2.0), [-0.12, 0.0, 0.08],
                                                   [-0.03, 0.0, 0.02], [-0.3, 0.0, 0.3]), (16.4, -10.2, 2, 0, 0), (3.0, 2.4)),
        ("deg_header_deg_lags", "E", "deg", (np.arange(9.0, 26.0, 2.0) / 3600.0, np.arange(-17.0, 0.0, 2.0) / 3600.0,
                                             [-2e-5, 0.0, 2e-5], [0.0, 1e-5], [0.0, 0.3]), (17.1 / 3600, -9.3 / 3600, 0, 1, 1),
         (3.0 / 3600, 2.8 / 3600)),
        ("arcsec_peak_at_an_edge", "A", "arcsec", (np.arange(9.0, 26.0, 2.0), np.arange(-17.0, 0.0, 2.0), [0.0, 0.05],
                                                   [-0.04, 0.0], [0.3]), (24.6, -16.4, 1, 0, 0), (2.2, 2.0)),
    ]
    for k, (name, sc, unit, axes, peak, widths) in enumerate(cases):
        rng = np.random.default_rng(94000 + k)
        axes = [np.asarray(a, dtype=np.float64) for a in axes]
        corr = make_map(rng, axes, peak, widths)
        res = AlignmentResults(corr=corr.copy(), lag_crval1=axes[0], lag_crval2=axes[1], lag_cdelt1=axes[2],
                               lag_cdelt2=axes[3], lag_crota=axes[4], unit_lag=unit, image_to_align_path=paths[sc],
                               image_to_align_window=-1)
        hdr = res.return_corrected_header(window=-1)
        out = os.path.join(tmp, name + "_corrected.fits")
        res.write_corrected_fits(window_list_to_apply_shift=[-1], path_to_l3_output=out)
        with fits.open(out) as hdul:
            written = {c: M.jsonable(hdul[-1].header[c]) for c in CARDS}
        ARR[f"case/{name}/corr"] = corr
        META["cases"][name] = {
            "scene": sc, "unit_lag": unit, "axes": [a.tolist() for a in axes],
            "max_index": [int(v) for v in res.max_index], "shift_pixels": M.jsonable(list(res.shift_pixels)),
            "shift_arcsec": M.jsonable(list(res.shift_arcsec)),
            "parameters_alignment_arcsec": {c: M.jsonable(v) for c, v in res.parameters_alignment_arcsec.items()},
            "corrected_header": {c: M.jsonable(hdr[c]) for c in CARDS}, "written_header": written}
        print(f"{name:26s} max_index {tuple(int(v) for v in res.max_index)} shift_arcsec {tuple(float(v) for v in res.shift_arcsec)}",
              flush=True)
    import astropy
    import scipy
    META["interpreter"] = {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                           "astropy": astropy.__version__}
    np.savez_compressed(os.path.join(HERE, "results_5d_golden.npz"), **ARR)
    with open(os.path.join(HERE, "results_5d_golden.json"), "w") as f:
        json.dump(META, f, indent=1, sort_keys=True)
    print("wrote results_5d_golden.npz / .json:", len(META["cases"]), "cases")


if __name__ == "__main__":
    main()
