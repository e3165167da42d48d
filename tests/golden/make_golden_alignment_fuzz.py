#!/opt/conda/bin/python3.9
"""
Golden-vector generator, random family: the REFERENCE's own `euispice_coreg.hdrshift.alignment.Alignment`
(`hdrshift/alignment.py:144-399, 401-468, 509-578, 613-797`) on 12 seeded random scenes x 3 calls each -- rolled and
off-centre reference images with unequal CDELT, rectangular images to align in arcsec or degrees, random pointing
errors, NaN fractions, Carrington grids, lag sets over CRVAL / CROTA / CDELT1 / solar radius, spline orders 1-3, serial
and parallel branches, thresholds.  Same file layout as `make_golden_alignment.py` (whose writer / runner it imports):

    tests/golden/alignment_fuzz_golden.npz    images (float32) and the reference's correlation maps
    tests/golden/alignment_fuzz_golden.json   headers as astropy read them back + the calls made

Where `alignment_golden` walks the quirk ledger case by case, this family checks that nothing depends on the three
hand-made scenes: every number below comes out of `np.random.default_rng(SEED + k)`.

Run (build container only; /root/reference must exist; about two minutes):
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_alignment_fuzz.py
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
SEED = 77000
N_SCENES = 12


def random_scene(rng):
    """A pair rendered from one set of blobs through the TRUE headers; the image to align is handed out with a wrong
    CRVAL / CROTA.  Unlike synthetic.make_scene: reference image rolled, off-centre CRPIX, unequal CDELT on both."""
    sny, snx = int(rng.integers(44, 84)), int(rng.integers(44, 84))
    lny, lnx = int(rng.integers(104, 136)), int(rng.integers(104, 136))
    unit = "deg" if rng.random() < 0.3 else "arcsec"
    u = {"arcsec": 1.0, "deg": 1.0 / 3600.0}[unit]
    fov = 2048 * 0.492
    cd1 = fov / snx * rng.uniform(0.9, 1.1)
    cd2 = fov / sny * rng.uniform(0.9, 1.1)
    true_crval = (-310.0 + rng.uniform(-60, 60), 420.0 + rng.uniform(-60, 60))
    err = (rng.uniform(-25, 25), rng.uniform(-25, 25), rng.uniform(-0.5, 0.5))
    crota = rng.uniform(-8.0, 8.0)
    crpix = ((snx + 1) / 2.0 + rng.uniform(-6, 6), (sny + 1) / 2.0 + rng.uniform(-6, 6))
    h_true = synthetic._header(snx, sny, crpix[0], crpix[1], true_crval[0] * u, true_crval[1] * u, cd1 * u, cd2 * u,
                               crota + err[2], unit=unit)
    h_small = synthetic._header(snx, sny, crpix[0], crpix[1], (true_crval[0] - err[0]) * u, (true_crval[1] - err[1]) * u,
                                cd1 * u, cd2 * u, crota, unit=unit)
    lcd = 3072 * 4.44 / max(lnx, lny)
    h_large = synthetic._header(lnx, lny, (lnx + 1) / 2.0 + rng.uniform(-8, 8), (lny + 1) / 2.0 + rng.uniform(-8, 8),
                                rng.uniform(-40, 40), rng.uniform(-40, 40), lcd * rng.uniform(0.95, 1.05),
                                lcd * rng.uniform(0.95, 1.05), rng.uniform(-4.0, 4.0), wavelnth=174,
                                date="2022-03-17T09:50:45.281")
    n_blobs = 110
    half = 0.5 * max(cd1 * snx, cd2 * sny) + 150.0
    blobs = np.empty((n_blobs, 4))
    blobs[:, 0] = true_crval[0] + rng.uniform(-half, half, n_blobs)
    blobs[:, 1] = true_crval[1] + rng.uniform(-half, half, n_blobs)
    blobs[:, 2] = rng.uniform(8.0, 60.0, n_blobs)
    blobs[:, 3] = np.exp(rng.uniform(np.log(50.0), np.log(3000.0), n_blobs))
    small = synthetic._render(h_true, blobs, 100.0, rng).astype(np.float32)
    large = synthetic._render(h_large, blobs, 100.0, rng).astype(np.float32)
    nan_frac = float(rng.choice([0.0, 0.004, 0.02]))
    if nan_frac > 0:
        small[rng.random(small.shape) < nan_frac] = np.nan
    return small, h_small, large, h_large, err, unit


def lag_axis(rng, centre, n, step):
    return [float(centre + step * (k - (n - 1) / 2.0) + 0.0) for k in range(n)]


def random_call(rng, err, unit_hdr, frame, parallel):
    """Constructor + call arguments as plain numbers.  Lags stay in arcsec (the reference converts them when the header is
    in degrees, alignment.py:819-837).  No d_cdelt2 != 0 and one solar radius only: those are the ledger's Q9 / Q10,
    pinned case by case in alignment_golden."""
    n1, n2 = int(rng.integers(3, 6)), int(rng.integers(3, 6))
    off = (rng.uniform(-2, 2), rng.uniform(-2, 2))  # the window is not centred on the truth
    ctor = dict(lag_crval1=lag_axis(rng, round(err[0]) + off[0], n1, float(rng.choice([2.0, 3.5, 5.0]))),
                lag_crval2=lag_axis(rng, round(err[1]) + off[1], n2, float(rng.choice([2.0, 3.5, 5.0]))),
                lag_cdelt1=None, lag_cdelt2=None, lag_crota=None, parallelism=bool(parallel))
    r = rng.random()
    if r < 0.55:
        ctor["lag_crota"] = sorted({0.0, round(float(err[2]), 2), round(float(rng.uniform(-0.6, 0.6)), 2)})[: int(rng.integers(1, 4))]
    if rng.random() < 0.3:
        ctor["lag_cdelt1"] = [-0.04, 0.0, 0.03][: int(rng.integers(2, 4))]
        ctor["lag_cdelt2"] = [0.0]
        if ctor["lag_crota"] is None:
            ctor["lag_crota"] = [0.0]
    ctor["reprojection_order"] = int(rng.choice([1, 2, 2, 3]))
    if rng.random() < 0.3:
        ctor["small_fov_value_min"] = 130.0
    if rng.random() < 0.2:
        ctor["small_fov_value_max"] = 1500.0
    if parallel:
        ctor["counts_cpu_max"] = int(rng.integers(2, 5))
    ck = {}
    if frame == "carrington":
        if rng.random() < 0.4:
            ctor["lag_solar_r"] = [round(float(rng.uniform(1.0, 1.03)), 4)]
        ck = {"lonlims": [226.0 + float(rng.integers(0, 4)), 260.0 + float(rng.integers(0, 4))],
              "latlims": [-13.0 + float(rng.integers(0, 3)), 21.0 + float(rng.integers(0, 3))],
              "shape": [int(rng.integers(48, 80)), int(rng.integers(48, 80))]}
    return ctor, ck


def reference_results(entry, corr, ctor):
    """The reference's own `AlignmentResults` on the map it returned (AlignmentResults.py:24-101 argmax and lag
    bookkeeping, :218-341 the sub-lag Gaussian fit through scipy 1.7.1's curve_fit): what `return_type='AlignmentResults'`
    hands the user for this call.  An all-NaN map, or a fit the reference cannot make, is recorded as such."""
    from euispice_coreg.hdrshift.AlignmentResults import AlignmentResults
    arr = {k: (None if ctor.get(k) is None else np.asarray(ctor[k], dtype=np.float64))
           for k in ("lag_crval1", "lag_crval2", "lag_cdelt1", "lag_cdelt2", "lag_crota")}
    try:
        res = AlignmentResults(corr=np.array(corr), unit_lag="arcsec", **arr)
        entry["results"] = {
            "max_index": [int(v) for v in res.max_index], "shift_pixels": M.jsonable(list(res.shift_pixels)),
            "shift_arcsec": M.jsonable(list(res.shift_arcsec)),
            "parameters_alignment_arcsec": {k: M.jsonable(v) for k, v in res.parameters_alignment_arcsec.items()}}
    except Exception as e:  # noqa: BLE001 -- what it raises IS the fixture
        entry["results_raises"] = type(e).__name__
        entry["results_message"] = str(e)[:160]


def main():
    tmp = tempfile.mkdtemp(prefix="golden_alignment_fuzz_")
    M.ARR.clear()
    M.META.update(scenes={}, cases={}, interpreter={})
    plan = [("helioprojective", True), ("carrington", False), ("helioprojective", False), ("carrington", True)]
    for k in range(N_SCENES):
        rng = np.random.default_rng(SEED + k)
        small, hs, large, hl, err, unit = random_scene(rng)
        name = f"S{k:02d}"
        paths = M.write_pair(tmp, name, small, hs, large, hl)
        M.META["scenes"][name]["truth"] = [float(v) for v in err]
        for j in range(3):
            frame, par = plan[(k + j) % 4]
            if unit == "deg" and frame == "carrington" and k != 1:
                # rectify.py reads CDELT / CRVAL as arcsec whatever CUNIT says: a header in degrees gives an all-NaN
                # map (kept for scene S01, both branches); the other scenes in degrees go through the TAN chain
                frame = "helioprojective"
            ctor, ck = random_call(rng, err, unit, frame, par)
            e_name = f"{name}_{j}_{frame[:5]}_{'par' if par else 'ser'}_o{ctor['reprojection_order']}"
            e = M.run_case(e_name, name, paths, ctor, frame, ck)
            assert "raises" not in e, e
            reference_results(e, M.ARR[f"case/{e_name}/corr"], ctor)
    import astropy
    import scipy
    M.META["interpreter"] = {"python": sys.version.split()[0], "numpy": np.__version__, "scipy": scipy.__version__,
                             "astropy": astropy.__version__, "seed": SEED}
    dst = os.path.join(HERE, "alignment_fuzz_golden.npz")
    np.savez_compressed(dst, **M.ARR)
    with open(os.path.join(HERE, "alignment_fuzz_golden.json"), "w") as f:
        json.dump(M.META, f, indent=1, sort_keys=True)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(M.META["cases"]), "cases")


if __name__ == "__main__":
    main()
