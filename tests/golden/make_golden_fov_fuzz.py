#!/opt/conda/bin/python3.9
"""
Golden-vector generator, random family for `fov_limits` / `remove_fov_limits`: the REFERENCE's own
`euispice_coreg.hdrshift.alignment.Alignment.align_using_helioprojective(fov_limits=..., remove_fov_limits=...)`
(`hdrshift/alignment.py:844-874` box set to NaN, `:1082-1127` re-grid of the image to align on a regular sub-FOV grid --
which REPLACES its header: CRPIX at the grid's mid point, CDELT from the mean pixel size, PCi_j = identity, CROTA = 0 --
`utils/Util.py:PlotFits.build_regular_grid`) on 8 seeded random scenes x 2 calls: rolled off-centre reference images,
rectangular images to align in arcsec or degrees, random boxes (a sub-FOV of 50-85 % per axis, a removed box of 15-35 %,
or both), CRVAL and CROTA lags, orders 1-3, both branches.  `alignment_golden` holds three hand-made calls with limits;
this family checks that agreement does not hang on them.  Same layout, writer and runner as `make_golden_alignment.py`;
scenes from `make_golden_alignment_fuzz.random_scene`.

    tests/golden/fov_fuzz_golden.npz / .json

Run (build container only; about a minute):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_fov_fuzz.py
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_alignment_fuzz as Fz  # noqa: E402  (imports make_golden_alignment, which loads the reference)

import numpy as np  # noqa: E402

M = Fz.M
SEED = 93000
N_SCENES = 8


def box(rng, centre, half, frac_lo, frac_hi):
    """[[lon0, lon1], [lat0, lat1]] in arcsec: a box of frac x the FOV per axis, placed at random inside it."""
    out = []
    for c, h in zip(centre, half):
        w = h * rng.uniform(frac_lo, frac_hi)
        mid = c + rng.uniform(-(h - w), h - w) * 0.8
        out.append([round(float(mid - w), 1), round(float(mid + w), 1)])
    return out


def main():
    tmp = tempfile.mkdtemp(prefix="golden_fov_fuzz_")
    M.ARR.clear()
    M.META.update(scenes={}, cases={}, interpreter={})
    for k in range(N_SCENES):
        rng = np.random.default_rng(SEED + k)
        small, hs, large, hl, err, unit = Fz.random_scene(rng)
        name = f"V{k:02d}"
        paths = M.write_pair(tmp, name, small, hs, large, hl)
        u = {"arcsec": 1.0, "deg": 3600.0}[unit]
        centre = (hs["CRVAL1"] * u, hs["CRVAL2"] * u)
        half = (0.5 * abs(hs["CDELT1"]) * u * hs["NAXIS1"], 0.5 * abs(hs["CDELT2"]) * u * hs["NAXIS2"])
        for j in range(2):
            par = bool((k + j) % 2)
            ctor, _ = Fz.random_call(rng, err, unit, "helioprojective", par)
            ctor["lag_cdelt1"] = ctor["lag_cdelt2"] = None
            kind = ("fov", "remove", "both")[(k + 2 * j) % 3]
            ck = {"limits_unit": "arcsec"}
            if kind in ("fov", "both"):
                ck["fov_limits"] = box(rng, centre, half, 0.5, 0.85)
            if kind in ("remove", "both"):
                ck["remove_fov_limits"] = box(rng, centre, half, 0.15, 0.35)
            e_name = f"{name}_{j}_{kind}_{'par' if par else 'ser'}_o{ctor['reprojection_order']}"
            M.run_case(e_name, name, paths, ctor, "helioprojective", ck)
    import astropy
    import scipy
    M.META["interpreter"] = {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                             "astropy": astropy.__version__, "seed": SEED}
    dst = os.path.join(HERE, "fov_fuzz_golden.npz")
    np.savez_compressed(dst, **M.ARR)
    with open(os.path.join(HERE, "fov_fuzz_golden.json"), "w") as f:
        json.dump(M.META, f, indent=1, sort_keys=True)
    n_raise = sum("raises" in c for c in M.META["cases"].values())
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(M.META["cases"]), "cases,", n_raise, "raise")


if __name__ == "__main__":
    main()
