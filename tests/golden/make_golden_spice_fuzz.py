#!/opt/conda/bin/python3.9
"""
Golden-vector generator, random family for the SPICE caller: the REFERENCE's own
`euispice_coreg.hdrshift.alignment_spice.AlignmentSpice.align_using_helioprojective` (`alignment_spice.py:66-120`,
`:189-221` _extract_spice_data_header, `:223-248` _correct_solar_rotation, `:250-323` _prepare_spice_from_l2;
`utils/Util.py:431-455` slit_pxl / vertical_edges_limits) on 8 seeded random SPICE-L2-like windows: raster size,
detector (SW / LW), NBIN2, PXBEG2, slit-time coupling PC4_1, NaN voxels and fully-NaN spectra in the cube, and the
options of the call (wavelength interval, `cut_from_center`, `sub_fov_window`, `extend_pixel_size`, CROTA lags, serial /
parallel branch) all come out of `np.random.default_rng(SEED + k)`.

    tests/golden/spice_fuzz_golden.npz    the factors each cube is rebuilt from + its NaN voxels, reference images,
                                          the reference's prepared 2-D images and correlation maps
    tests/golden/spice_fuzz_golden.json   4-D headers as astropy read them back, calls, the 2-D header cards composed

Run (build container only; about half a minute):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_spice_fuzz.py
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_callers as M  # noqa: E402  (loads the reference through _reference_loader)

import numpy as np  # noqa: E402
import astropy.units as u  # noqa: E402
from astropy.io import fits  # noqa: E402

synthetic = M.synthetic
SEED = 88000
N_SCENES = 8


def build_cube(image, profile, nan_voxels, nan_spectra):
    """float32 [1, nw, ny, nx]; the tests rebuild it the same way (IEEE multiplication + float32 rounding)."""
    cube = (image[None, :, :] * profile[:, None, None])[None].astype(np.float32)
    for w, y, x in nan_voxels:
        cube[0, w, y, x] = np.nan
    for y, x in nan_spectra:
        cube[0, :, y, x] = np.nan
    return cube


def main():
    tmp = tempfile.mkdtemp(prefix="golden_spice_fuzz_")
    ARR, META = {}, {"scenes": {}, "interpreter": {}}
    for k in range(N_SCENES):
        rng = np.random.default_rng(SEED + k)
        nx, ny, nw = int(rng.integers(30, 54)), int(rng.integers(112, 168)), int(rng.integers(6, 11))
        err = (float(rng.uniform(-30, 30)), float(rng.uniform(-30, 30)), 0.0)
        cube, h4, large, hl, truth = synthetic.make_spice_l2(nx=nx, ny=ny, nw=nw, large_n=128, seed=SEED + 100 + k,
                                                             pointing_error=err, n_blobs=120)
        nbin = int(rng.choice([2, 4]))
        h4.update(DETECTOR=str(rng.choice(["SW", "LW"])), NBIN2=nbin,
                  PXBEG2=int(rng.integers(180, 214)) if nbin == 4 else int(rng.integers(170, 200)),
                  PC4_1=float(-rng.uniform(5.0, 60.0)))
        image = np.asarray(truth["image"], dtype=np.float64)
        profile = np.asarray(truth["profile"], dtype=np.float64)
        n_vox = int(rng.integers(0, 60))
        nan_voxels = np.stack([rng.integers(0, nw, n_vox), rng.integers(0, ny, n_vox), rng.integers(0, nx, n_vox)],
                              axis=1).astype(np.int64)
        n_spec = int(rng.integers(0, 12))  # every wavelength NaN there: np.nansum gives 0.0, a VALID sample
        nan_spectra = np.stack([rng.integers(0, ny, n_spec), rng.integers(0, nx, n_spec)], axis=1).astype(np.int64)
        cube = build_cube(image, profile, nan_voxels, nan_spectra)
        large32 = np.asarray(large, dtype=np.float32)
        name = f"P{k:02d}"
        p_spice = os.path.join(tmp, f"solo_L2_spice-n-ras_202203{10 + k}T094045_V01.fits")
        fits.HDUList([fits.PrimaryHDU(data=cube, header=M.to_header(h4))]).writeto(p_spice, overwrite=True)
        p_large = os.path.join(tmp, f"solo_L2_eui-fsi174-image_ref{k}.fits")
        fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=large32, header=M.to_header(hl))]).writeto(p_large,
                                                                                                     overwrite=True)
        for key, v in (("image", image), ("profile", profile), ("nan_voxels", nan_voxels), ("nan_spectra", nan_spectra),
                       ("large", large32)):
            ARR[f"{name}/{key}"] = v
        with fits.open(p_spice) as f:
            hdr4d = M.cards(f[0].header)
        with fits.open(p_large) as f:
            hdr_large = M.cards(f[-1].header)

        # ---- the call
        n1, n2 = int(rng.integers(4, 7)), int(rng.integers(4, 7))
        s1, s2 = float(rng.choice([3.0, 4.0, 6.0])), float(rng.choice([2.0, 3.0, 4.0]))
        ctor = dict(lag_crval1=[round(err[0]) + s1 * (i - (n1 - 1) / 2.0) + 0.0 for i in range(n1)],
                    lag_crval2=[round(err[1]) + s2 * (i - (n2 - 1) / 2.0) + 0.0 for i in range(n2)],
                    lag_crota=[0.0] if rng.random() < 0.5 else [-0.4, 0.0, 0.5][: int(rng.integers(2, 4))],
                    parallelism=bool(rng.random() < 0.25))
        if ctor["parallelism"]:
            ctor["counts_cpu_max"] = 3
        call_kwargs, quantities = {}, {}
        if rng.random() < 0.5:
            a = int(rng.integers(0, nw // 2))
            b = int(rng.integers(a + 1, nw))
            wave = (h4["CRVAL3"] + h4["CDELT3"] * (np.arange(nw) + 1 - h4["CRPIX3"])) * 10.0  # angstrom
            quantities["wavelength_interval_to_sum"] = [float(wave[a] - 1e-4), float(wave[b] + 1e-4)]
        if rng.random() < 0.45:
            call_kwargs["cut_from_center"] = int(rng.integers(nx // 2, nx - 2))
        if rng.random() < 0.45:
            c1, c2 = h4["CRVAL1"], h4["CRVAL2"]
            quantities["sub_fov_window"] = [float(c1 - rng.uniform(30, 80)), float(c1 + rng.uniform(30, 80)),
                                            float(c2 - rng.uniform(30, 70)), float(c2 + rng.uniform(30, 70))]
        if rng.random() < 0.35:
            call_kwargs["extend_pixel_size"] = True
        kw = {kk: (np.asarray(v, dtype=np.float64) if kk.startswith("lag_") else v) for kk, v in ctor.items()}
        if "wavelength_interval_to_sum" in quantities:
            kw["wavelength_interval_to_sum"] = [v * u.angstrom for v in quantities["wavelength_interval_to_sum"]]
        if "sub_fov_window" in quantities:
            kw["sub_fov_window"] = [v * u.arcsec for v in quantities["sub_fov_window"]]
        A = M.AlignmentSpice(large_fov_known_pointing=p_large, small_fov_to_correct=p_spice, small_fov_window=0,
                             large_fov_window=-1, **kw)
        corr = A.align_using_helioprojective(return_type="corr", **call_kwargs)
        ARR[f"{name}/corr"] = np.asarray(corr, dtype=np.float64)
        entry = {"hdr4d": hdr4d, "hdr_large": hdr_large, "truth": list(err), "ctor": M.jsonable(ctor),
                 "call_kwargs": M.jsonable(call_kwargs), "quantities": M.jsonable(quantities),
                 "shape": list(corr.shape), "hdr_small": M.cards(A.hdr_small),
                 "file": os.path.basename(p_spice)}
        if not ctor["parallelism"]:  # the parallel branch deletes the prepared image (alignment.py:660-665)
            ARR[f"{name}/data_small"] = np.asarray(A.data_small, dtype=np.float64)
            entry["prepared_nan"] = int(np.isnan(A.data_small).sum())
            entry["prepared_zero"] = int((A.data_small == 0.0).sum())
        META["scenes"][name] = entry
        print(f"{name} nx {nx} ny {ny} nw {nw} {h4['DETECTOR']} nbin {nbin} corr {tuple(corr.shape)} "
              f"nan {int(np.isnan(corr).sum())} max {np.nanmax(corr):.6f} argmax "
              f"{np.unravel_index(np.nanargmax(corr), corr.shape)[:2]} opts {sorted(call_kwargs)} {sorted(quantities)} "
              f"par {ctor['parallelism']}", flush=True)
    import astropy
    import scipy
    META["interpreter"] = {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                           "astropy": astropy.__version__, "seed": SEED}
    dst = os.path.join(HERE, "spice_fuzz_golden.npz")
    np.savez_compressed(dst, **ARR)
    with open(os.path.join(HERE, "spice_fuzz_golden.json"), "w") as f:
        json.dump(META, f, indent=1, sort_keys=True)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(META["scenes"]), "scenes")


if __name__ == "__main__":
    main()
