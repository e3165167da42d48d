"""Re-run one case of tests/deep_fuzz.py with diagnostics.  usage: python tests/fuzz_one.py <seed> [scale]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from euispice_coreg_amd import _lib
    from tests import helpers as H
    from tests.test_gpu_fuzz import _random_case
    seed = int(sys.argv[1])
    scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    h = _lib.CoregHandle(-1)
    small, hs, large, hl, lags, rng = _random_case(seed, scale)
    order = int(rng.choice([1, 2]))
    sem = str(rng.choice(["intended", "reference"]))
    lags = list(lags)
    if rng.integers(0, 3) == 0 and sem == "intended":
        lags[3] = [0.0, -0.02]
    frame = "carrington" if seed % 2 == 0 else "helio"
    print(f"seed {seed} frame {frame} order {order} sem {sem} small {small.shape} nan {np.isnan(small).mean():.4f} "
          f"f32exact {np.array_equal(small[np.isfinite(small)], small[np.isfinite(small)].astype(np.float32))}")
    print("lags", [None if l is None else np.asarray(l).round(3).tolist() for l in lags])
    print("hdr small", {k: hs[k] for k in ("CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "CROTA", "CRPIX1", "CRPIX2", "NAXIS1", "NAXIS2")})
    got = {}
    for clean in (1, 0):
        h.set_option("clean_path", clean)
        if frame == "carrington":
            lon0 = float(rng.choice([228.0, 200.0, 150.0])) if clean == 1 else lon0
            if clean == 1:
                lonlims = (lon0, lon0 + float(rng.choice([34.0, 100.0, 220.0])))
                latlims = (-12.0 - float(rng.choice([0, 60])), 22.0)
                shape = (int(rng.integers(20, 70)) * scale, int(rng.integers(20, 70)) * scale)
                solar_r = float(rng.choice([1.004, 1.0, 1.02]))
                print("grid", lonlims, latlims, shape, solar_r)
                want = H.oracle_carrington(small, hs, large, hl, lags, shape, lonlims, latlims, order=order,
                                           solar_r=(solar_r,), cdelt_semantics=sem)
            got[clean] = H.gpu_carrington(h, small, hs, large, hl, lags, shape, lonlims, latlims, order=order,
                                          solar_r=solar_r, cdelt_semantics=0 if sem == "intended" else 1)
        else:
            if clean == 1:
                serial = bool(rng.integers(0, 2))
                print("serial", serial)
                want = H.oracle_helio(small, hs, large, hl, lags, order=order, parallelism=not serial,
                                      cdelt_semantics=sem)
            got[clean] = H.gpu_helio(h, small, hs, large, hl, lags, order=order, serial_semantics=serial,
                                     cdelt_semantics=0 if sem == "intended" else 1)
        print("clean_path", clean, "visits", h.last_visit_counts(), "stats", {k: v for k, v in h.last_stats().items() if k.startswith("n_")})
    d = np.abs(got[1] - want)
    print("nan pattern equal:", np.array_equal(np.isnan(got[1]), np.isnan(want)))
    idx = np.unravel_index(np.nanargmax(d), d.shape)
    print("max |dcorr|", np.nanmax(d), "at", idx, "gpu", got[1][idx], "oracle", want[idx], "gpu(clean_path=0)", got[0][idx])
    print("max |clean - masked|", np.nanmax(np.abs(got[1] - got[0])))
    worst = np.argsort(np.nan_to_num(d.ravel()))[::-1][:6]
    for w in worst:
        i = np.unravel_index(w, d.shape)
        print("  ", i, d[i], got[1][i], want[i])
    use_lds0 = None
    h.set_option("use_lds", 0)
    try:
        if frame == "carrington":
            use_lds0 = H.gpu_carrington(h, small, hs, large, hl, lags, shape, lonlims, latlims, order=order, solar_r=solar_r,
                                        cdelt_semantics=0 if sem == "intended" else 1)
        else:
            use_lds0 = H.gpu_helio(h, small, hs, large, hl, lags, order=order, serial_semantics=serial,
                                   cdelt_semantics=0 if sem == "intended" else 1)
    finally:
        h.set_option("use_lds", 1)
    print("global-memory path at the worst lag-point:", use_lds0[idx], " max |lds - global|", np.nanmax(np.abs(use_lds0 - got[1])))


if __name__ == "__main__":
    main()
