#!/opt/conda/bin/python3.9
"""
Golden-vector generator, random family for the synthetic-raster builder: the REFERENCE's own
`euispice_coreg.synras.map_builder.SPICEComposedMapBuilder.process` (`map_builder.py:57-79`, `:87-214`, `:249-294`) on
four of the random SPICE L2 windows of `spice_fuzz_golden` (raster size, detector, NBIN2, slit-time coupling PC4_1 all
random there) with imager sequences whose start, cadence and length come out of `np.random.default_rng(SEED + k)` --
rasters lasting from a few minutes to over an hour, 3 to 16 imager frames -- with and without
`keep_original_imager_pixel_size`.

    tests/golden/synras_fuzz_golden.npz    the synthetic rasters the reference wrote
    tests/golden/synras_fuzz_golden.json   the sequences (start, cadence, frames; headers as astropy read them back),
                                           the composed headers, the imager frame taken for every raster column

The inputs are rebuilt by the tests from `spice_fuzz_golden` (cube factors, reference image) and
`synthetic.make_imager_sequence` (IEEE multiplication + float32 rounding: deterministic).

Run (build container only, after make_golden_spice_fuzz.py; seconds):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_synras_fuzz.py
"""
import datetime as dt
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_callers as M  # noqa: E402  (loads the reference through _reference_loader)
from make_golden_spice_fuzz import build_cube  # noqa: E402

import numpy as np  # noqa: E402
import astropy.units as u  # noqa: E402
from astropy.io import fits  # noqa: E402

SEED = 66000
WINDOWS = ["P00", "P02", "P05", "P07"]


def main():
    tmp = tempfile.mkdtemp(prefix="golden_synras_fuzz_")
    g = np.load(os.path.join(HERE, "spice_fuzz_golden.npz"))
    with open(os.path.join(HERE, "spice_fuzz_golden.json")) as f:
        sp = json.load(f)["scenes"]
    ARR, META = {}, {"cases": {}, "interpreter": {}}
    for k, name in enumerate(WINDOWS):
        rng = np.random.default_rng(SEED + k)
        h4, hl = dict(sp[name]["hdr4d"]), dict(sp[name]["hdr_large"])
        cube = build_cube(g[f"{name}/image"], g[f"{name}/profile"], g[f"{name}/nan_voxels"], g[f"{name}/nan_spectra"])
        d = os.path.join(tmp, name)
        os.makedirs(d)
        p_spice = os.path.join(d, sp[name]["file"])
        fits.HDUList([fits.PrimaryHDU(data=cube, header=M.to_header(h4))]).writeto(p_spice, overwrite=True)
        # the raster's time span from its header (time axis: CRVAL4 + CDELT4 * PC4_1 * (x + 1 - CRPIX1) seconds after DATEREF)
        nx = h4["NAXIS1"]
        t_col = h4["CRVAL4"] + h4["CDELT4"] * h4["PC4_1"] * (np.arange(nx) + 1 - h4["CRPIX1"])
        cadence = float(rng.choice([120.0, 300.0, 450.0]))
        t0 = dt.datetime.strptime(h4["DATEREF"], "%Y-%m-%dT%H:%M:%S.%f") + dt.timedelta(
            seconds=float(t_col.min()) - float(rng.uniform(20.0, 0.45 * cadence)))
        n_frames = int(np.ceil((t_col.max() - t_col.min()) / cadence)) + 2
        start = t0.strftime("%Y-%m-%dT%H:%M:%S.%f")[:-3]
        frames = M.synthetic.make_imager_sequence(g[f"{name}/large"].astype(np.float64), hl, start=start,
                                                  cadence_s=cadence, n_frames=n_frames)
        paths, imager_headers = [], []
        for j, (img, h) in enumerate(frames):
            p = os.path.join(d, f"solo_L2_eui-fsi174-image_{j:02d}.fits")
            fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=img, header=M.to_header(h))]).writeto(p, overwrite=True)
            paths.append(p)
            with fits.open(p) as f:
                imager_headers.append(M.cards(f[-1].header))
        keep = bool(k % 2)
        threshold = cadence
        C = M.SPICEComposedMapBuilder(path_to_spectro=p_spice, list_imager_paths=paths,
                                      threshold_time=u.Quantity(threshold, "s"), window_imager=-1, window_spectro=0)
        out = C.process(folder_path_output=d, basename_output="synras.fits", print_filename=False,
                        return_synras_name=True, **({"keep_original_imager_pixel_size": True} if keep else {}))
        with fits.open(out) as f:
            ARR[f"{name}/raster"] = np.asarray(f[0].data, dtype=np.float64)
            META["cases"][name] = {
                "start": start, "cadence_s": cadence, "n_frames": n_frames, "threshold_time": threshold,
                "kwargs": {"keep_original_imager_pixel_size": True} if keep else {},
                "imager_headers": imager_headers, "header": M.cards(f[0].header), "shape": list(f[0].data.shape),
                "frame_of_column": [int(np.argmin([abs((dd - t).to("s").value) for t in C.dates]))
                                    for dd in C.dates_selected]}
        c = META["cases"][name]
        print(f"{name} raster {c['shape']} span {t_col.max() - t_col.min():.0f} s cadence {cadence:.0f} frames {n_frames} "
              f"used {sorted(set(c['frame_of_column']))} keep {keep} nan {int(np.isnan(ARR[name + '/raster']).sum())}",
              flush=True)
    import astropy
    import scipy
    META["interpreter"] = {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                           "astropy": astropy.__version__, "seed": SEED}
    dst = os.path.join(HERE, "synras_fuzz_golden.npz")
    np.savez_compressed(dst, **ARR)
    with open(os.path.join(HERE, "synras_fuzz_golden.json"), "w") as f:
        json.dump(META, f, indent=1, sort_keys=True)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
