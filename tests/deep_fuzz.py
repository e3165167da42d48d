"""Long randomised GPU-vs-oracle parity run (the generator of tests/test_gpu_fuzz.py over many more seeds, plus
method='residus', CDELT2 lags, both CDELT semantics, degrees headers; round 4: the compile-time cubic kernel with
`orders` = 1,2,3, and -- Carrington cases with several (cdelt, crota) combinations -- the sweep stitched from two
runs of combinations, the one-shot "combo_begin" / "combo_end" option of a multi-GPU share).
With `forms` = 1 the images no longer go up as arrays only: per case, at random, the image to align and / or the reference
image are written to disk first -- as a plain FITS data unit (float32 / float64 / scaled int16, uploaded raw and decoded
on the GPU) or tile-compressed (RICE_1, quantized float32 with a random dither method, tiling and seed; the compressed
bytes go up and the GPU decodes them) -- and the ORACLE gets the pixels the host readers decode from those files.
usage: python tests/deep_fuzz.py [n] [seed0] [scale] [orders, e.g. 1,2,3] [forms 0|1]"""
import os
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


_TMP = None
FORM_COUNT = {}


def through_a_file(img, hdr, rng, tag):
    """(what the library is handed, the pixels the oracle must see)"""
    import tempfile
    import numpy as np
    from euispice_coreg_amd.utils import fits_io
    global _TMP
    if _TMP is None:
        _TMP = tempfile.mkdtemp(prefix="coreg_fuzz_")
    kind = str(rng.choice(["array", "raw_f32", "raw_f64", "raw_i16", "rice_f32", "rice_i16"]))
    FORM_COUNT[kind] = FORM_COUNT.get(kind, 0) + 1
    if kind == "array":
        return img, img, kind
    p = os.path.join(_TMP, tag + ".fits")
    if kind == "raw_f32":
        fits_io.write_images(p, [(None, {}), (img.astype(np.float32), hdr)])
    elif kind == "raw_f64":
        fits_io.write_images(p, [(None, {}), (img.astype(np.float64), hdr)])
    elif kind in ("raw_i16", "rice_i16"):
        lo, hi = np.nanmin(img), np.nanmax(img)
        px = np.nan_to_num((img - lo) / max(hi - lo, 1e-30) * 60000.0 - 30000.0, nan=-32768.0).astype(np.int16)
        if kind == "raw_i16":
            fits_io.write_images(p, [(None, {}), (px, hdr)])
        else:
            fits_io.write_compressed_image(p, px, hdr, tile=None if rng.integers(0, 2) else (int(rng.integers(3, 40)), int(rng.integers(1, 9))))
    else:
        fits_io.write_compressed_image(p, img.astype(np.float32), hdr,
                                       quantize=str(rng.choice(["NO_DITHER", "SUBTRACTIVE_DITHER_1", "SUBTRACTIVE_DITHER_2"])),
                                       dither0=int(rng.integers(1, 10001)), scale=float(np.nanstd(img)) / float(rng.choice([16, 64, 1000])),
                                       tile=None if rng.integers(0, 2) else (int(rng.integers(3, 40)), int(rng.integers(1, 9))))
    up = fits_io.load_for_upload(p, -1)[0]
    return up, np.asarray(fits_io.read_image(p, -1)[0], dtype=np.float64), kind


def build_case(seed, scale, orders, forms):
    """Everything a case draws before its sweeps, in the order the random generator is consumed."""
    import numpy as np
    from tests.test_gpu_fuzz import _random_case
    small, hs, large, hl, lags, rng = _random_case(seed, scale)
    order = int(rng.choice(orders))
    sem = str(rng.choice(["intended", "reference"]))
    lags = list(lags)
    if rng.integers(0, 3) == 0 and sem == "intended":  # the reference dies on a CDELT2 lag (quirk Q2)
        lags[3] = [0.0, -0.02]
    frame = "carrington" if seed % 2 == 0 else "helio"
    small_up, large_up, used = small, large, ("array", "array")
    if forms:
        small_up, small, k1 = through_a_file(small, hs, rng, "small")
        large_up, large, k2 = through_a_file(large, hl, rng, "large")
        used = (k1, k2)
    c = dict(small=small, hs=hs, large=large, hl=hl, lags=lags, rng=rng, order=order, sem=sem, frame=frame,
             small_up=small_up, large_up=large_up, forms_used=used)
    if frame == "carrington":
        lon0 = float(rng.choice([228.0, 200.0, 150.0]))
        c["lonlims"] = (lon0, lon0 + float(rng.choice([34.0, 100.0, 220.0])))
        c["latlims"] = (-12.0 - float(rng.choice([0, 60])), 22.0)
        c["shape"] = (int(rng.integers(20, 70)) * scale, int(rng.integers(20, 70)) * scale)
        c["solar_r"] = float(rng.choice([1.004, 1.0, 1.02]))
    else:
        c["serial"] = bool(rng.integers(0, 2)) and not (forms and not isinstance(large_up, np.ndarray))
    return c


def main():
    from euispice_coreg_amd import _lib
    from tests import helpers as H
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # image / grid size multiplier
    orders = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2]
    forms = len(sys.argv) > 5 and sys.argv[5] == "1"
    import numpy as np
    n_split = 0
    h = _lib.CoregHandle(-1)
    bad = 0
    # tile visits of each case's LAST sweep launch; lag-points re-evaluated with centred sums (whole sweeps)
    kinds = {"visits": 0, "lds": 0, "interior": 0, "all_finite": 0, "refined_lag_points": 0, "flagged_not_refined": 0}
    t0 = time.time()
    for seed in range(seed0, seed0 + n):
        c = build_case(seed, scale, orders, forms)
        small, hs, large, hl, lags, rng = c["small"], c["hs"], c["large"], c["hl"], c["lags"], c["rng"]
        order, sem, frame, small_up, large_up = c["order"], c["sem"], c["frame"], c["small_up"], c["large_up"]
        try:
            if frame == "carrington":
                lonlims, latlims, shape, solar_r = c["lonlims"], c["latlims"], c["shape"], c["solar_r"]
                want = H.oracle_carrington(small, hs, large, hl, lags, shape, lonlims, latlims, order=order,
                                           solar_r=(solar_r,), cdelt_semantics=sem)
                got = H.gpu_carrington(h, small_up, hs, large_up, hl, lags, shape, lonlims, latlims, order=order,
                                       solar_r=solar_r, cdelt_semantics=0 if sem == "intended" else 1)
                ls = _lib.LagSet(*lags)
                inner = ls.shape[2] * ls.shape[3] * ls.shape[4]
                if inner > 1 and len(orders) > 2:  # the same map from two runs of combinations (a multi-GPU share each)
                    cut = int(rng.integers(1, inner))
                    parts = []
                    for c_lo, c_hi in ((0, cut), (cut, inner)):
                        h.set_option("combo_begin", c_lo)
                        h.set_option("combo_end", c_hi)
                        parts.append(H.gpu_carrington(h, small, hs, large, hl, lags, shape, lonlims, latlims, order=order,
                                                      solar_r=solar_r, cdelt_semantics=0 if sem == "intended" else 1,
                                                      prepare=False, lag_end=ls.shape[0] * ls.shape[1] * (c_hi - c_lo)
                                                      ).reshape(ls.shape[0], ls.shape[1], c_hi - c_lo))
                    stitched = np.concatenate(parts, axis=2).reshape(got.shape)
                    assert np.array_equal(np.isnan(stitched), np.isnan(got)), f"seed={seed}: combo runs, NaN pattern"
                    if np.isfinite(got).any():
                        assert np.nanmax(np.abs(stitched - got)) <= 1e-12, f"seed={seed}: combo runs differ from the whole"
                    n_split += 1
                tol = 1e-9
            else:
                serial = c["serial"]
                want = H.oracle_helio(small, hs, large, hl, lags, order=order, parallelism=not serial,
                                      cdelt_semantics=sem)
                got = H.gpu_helio(h, small_up, hs, large_up, hl, lags, order=order, serial_semantics=serial,
                                  cdelt_semantics=0 if sem == "intended" else 1)
                tol = 1e-7
            for k, v in h.last_visit_counts().items():
                kinds[k] += v
            H.assert_corr_close(got, want, tol, f"seed={seed} {frame}")
        except AssertionError as e:
            bad += 1
            print(f"FAIL seed={seed} frame={frame} order={order} sem={sem}: {e}", flush=True)
        if (seed - seed0) % 20 == 19:
            print(f"[deep_fuzz] {seed - seed0 + 1}/{n} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"[deep_fuzz] done: {n} cases (orders {orders}, {n_split} stitched from two combination runs), {bad} failures; "
          f"tile visits of the last launches: {kinds}" + (f"; image forms: {FORM_COUNT}" if forms else ""))
    import shutil
    if _TMP:
        shutil.rmtree(_TMP, ignore_errors=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
