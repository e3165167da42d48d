#!/opt/conda/bin/python3.9
"""
Golden-vector generator, random family for the PLATE-CARREE frame: the REFERENCE's own
`euispice_coreg.hdrshift.alignment.Alignment.align_using_initial_carrington` (`hdrshift/alignment.py:344-399`, sub-map in
both branches :649-651 / :765-767, `_shift_header` :401-468, `_extract_coordinates_pixels` :1038-1069 through astropy's
CAR projection) on 10 seeded random pairs of Carrington maps x 2 calls each -- rolled and unrolled maps, unequal and
negative pixel sizes, both hemispheres and the equator, an explicit LONPOLE, NaN fractions, lag sets in degrees over
CRVAL (through and around zero) and CROTA, spline orders 1-3, serial and parallel branches, thresholds.
`alignment_golden` holds two hand-made calls of this entry point; this family is the plate-carree counterpart of
`alignment_fuzz_golden` (same file layout, same writer / runner, imported from `make_golden_alignment.py`):

    tests/golden/car_fuzz_golden.npz    maps (float32) and the reference's correlation maps
    tests/golden/car_fuzz_golden.json   headers as astropy read them back + the calls made

A call the reference cannot finish (an explicit LONPOLE with a CRVAL2 lag across the equator: astropy raises inside the
worker) is recorded as such by the runner.

Run (build container only; /root/reference must exist; about a minute):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_car_fuzz.py
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_alignment as M  # noqa: E402  (loads the reference through _reference_loader)

import numpy as np  # noqa: E402

synthetic = M.synthetic
SEED = 91000
N_SCENES = 10


def random_scene(rng, k):
    ny, nx = int(rng.integers(44, 84)), int(rng.integers(44, 84))
    cd = (0.0101 * rng.uniform(0.8, 1.3), 0.0099 * rng.uniform(0.8, 1.3))
    crota = float(rng.choice([0.0, 0.0, 0.4, -2.5, 11.0]))
    small, hs, large, hl, truth = synthetic.make_car_scene(
        small_shape=(ny, nx), large_shape=(int(ny * 1.5), int(nx * 1.6)), seed=SEED + 100 + k, small_cdelt=cd, crota=crota,
        nan_frac=float(rng.choice([0.0, 0.004, 0.02])), explicit_lonpole=bool(k % 5 == 3))
    hs, hl = dict(hs), dict(hl)
    if rng.random() < 0.3:  # a map stored with longitude decreasing along x
        hs["CDELT1"] = -hs["CDELT1"]
        lam, rho = hs["CDELT2"] / hs["CDELT1"], np.deg2rad(hs["CROTA"])
        hs["PC1_2"], hs["PC2_1"] = float(-lam * np.sin(rho)), float(np.sin(rho) / lam)
        small = small[:, ::-1].copy()
    off = float(rng.choice([0.0, 0.0, 9.0, -21.0, 33.0]))
    if "LONPOLE" in hl:
        off = abs(off) + 0.5  # an explicit LONPOLE = 0 is only valid north of the equator
    hs["CRVAL2"] += off
    hl["CRVAL2"] += off
    return small, hs, large, hl, truth


def random_call(rng, truth, parallel, small):
    """Lags in DEGREES (unit_lag='deg': the maps are in degrees).  Half of the CRVAL axes pass through exactly zero --
    the lag-points whose border pixels and tap sets wcslib's rounding noise decides."""
    e1, e2 = truth["lag_crval1"], truth["lag_crval2"]

    def axis(e, through_zero):
        n, step = int(rng.integers(3, 6)), float(rng.choice([0.004, 0.0075, 0.011]))
        c = round(e / step) * step if through_zero else e + rng.uniform(-0.3, 0.3) * step
        a = [round(c + step * (j - (n - 1) // 2), 6) for j in range(n)]
        if through_zero and 0.0 not in a:
            a[int(np.argmin(np.abs(a)))] = 0.0
        return sorted(set(a))

    ctor = dict(lag_crval1=axis(e1, rng.random() < 0.5), lag_crval2=axis(e2, rng.random() < 0.5), lag_cdelt1=None,
                lag_cdelt2=None, lag_crota=None, unit_lag="deg", parallelism=bool(parallel),
                reprojection_order=int(rng.choice([1, 2, 2, 3])))
    if rng.random() < 0.5:
        ctor["lag_crota"] = sorted({0.0, round(float(rng.uniform(-0.6, 0.6)), 2)})
    if rng.random() < 0.3:  # the brightest 15 % of the map masked (these maps are a few large blobs: no fixed DN level fits)
        ctor["small_fov_value_max"] = round(float(np.nanpercentile(small, 85.0)), 1)
    if parallel:
        ctor["counts_cpu_max"] = int(rng.integers(2, 5))
    return ctor


def main():
    tmp = tempfile.mkdtemp(prefix="golden_car_fuzz_")
    M.ARR.clear()
    M.META.update(scenes={}, cases={}, interpreter={})
    for k in range(N_SCENES):
        rng = np.random.default_rng(SEED + k)
        small, hs, large, hl, truth = random_scene(rng, k)
        name = f"C{k:02d}"
        paths = M.write_pair(tmp, name, small, hs, large, hl)
        M.META["scenes"][name]["truth"] = [float(truth["lag_crval1"]), float(truth["lag_crval2"])]
        for j in range(2):
            par = bool((k + j) % 2)
            ctor = random_call(rng, truth, par, small)
            e_name = f"{name}_{j}_{'par' if par else 'ser'}_o{ctor['reprojection_order']}"
            M.run_case(e_name, name, paths, ctor, "initial_carrington")
    import astropy
    import scipy
    M.META["interpreter"] = {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                             "astropy": astropy.__version__, "seed": SEED}
    dst = os.path.join(HERE, "car_fuzz_golden.npz")
    np.savez_compressed(dst, **M.ARR)
    with open(os.path.join(HERE, "car_fuzz_golden.json"), "w") as f:
        json.dump(M.META, f, indent=1, sort_keys=True)
    n_raise = sum("raises" in c for c in M.META["cases"].values())
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(M.META["cases"]), "cases,", n_raise, "raise")


if __name__ == "__main__":
    main()
