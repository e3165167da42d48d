"""Throughput of the jitter-correction session on a synthetic 2048^2 series: images/s end to end
(FITS decode -> upload -> thresholds -> 100x100-lag Carrington sweep -> Gaussian fit -> corrected FITS written),
with the per-stage costs measured separately.  usage: python profiles/jitter_bench.py [n_frames] [n]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from euispice_coreg_amd import _lib, synthetic
    from euispice_coreg_amd.jitter_correction import jitter_correction_imagers
    from euispice_coreg_amd.utils import fits_io
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    lon, lat, shape = (228.0, 262.0), (-12.0, 22.0), (n, n)
    d = tempfile.mkdtemp(prefix="jitter_")
    t0 = time.time()
    frames, jit = synthetic.make_series(n_frames=n_frames, n=n, seed=3, n_blobs=300, jitter_sigma=1.5)
    paths = []
    for k, (img, hdr) in enumerate(frames):
        p = os.path.join(d, f"frame_{k:03d}.fits")
        fits_io.write_images(p, [(None, {}), (img, hdr)])
        paths.append(p)
    print(f"[jitter_bench] {n_frames} frames of {n}^2 rendered and written in {time.time() - t0:.1f} s", file=sys.stderr)

    kw = dict(lonlims=lon, latlims=lat, shape=shape, sublist_length=n_frames - 1, overlap=1, small_fov_value_max=2800.0)
    out = os.path.join(d, "out")
    jitter_correction_imagers(paths[:2], os.path.join(d, "warm"), **kw)  # warm-up: library, buffers, page cache
    t0 = time.perf_counter()
    done = jitter_correction_imagers(paths, out, **kw)
    dt = time.perf_counter() - t0
    err = []
    for idx, ref, res in done:
        err.append([res.shift_arcsec[0] - jit[idx, 0], res.shift_arcsec[1] - jit[idx, 1]])
    err = np.abs(np.array(err))

    # stage costs, one image, serial
    t = {}
    t0 = time.perf_counter(); img, hdr = fits_io.read_image(paths[1], -1); t["fits_decode_host_ms"] = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter(); raw, hdr = fits_io.load_for_upload(paths[1], -1); t["fits_open_raw_ms"] = 1e3 * (time.perf_counter() - t0)
    from euispice_coreg_amd.jitter_correction.jitter_correction import session_devices
    h = _lib.shared_handle(*session_devices(None, 1)[0])  # (a context the session has left its prepared reference in)
    t0 = time.perf_counter(); h.set_small(raw); h.threshold_small(None, 2800.0); t["upload_decode_threshold_ms"] = 1e3 * (time.perf_counter() - t0)
    grid = _lib.Grid(lon, lat, shape)
    lags = _lib.LagSet(np.arange(-5, 5, 0.1), np.arange(-5, 5, 0.1), None, None, None)
    t0 = time.perf_counter(); h.sweep_carrington(hdr, grid, 1.004, lags); t["sweep_call_ms"] = 1e3 * (time.perf_counter() - t0)
    st = h.last_stats()
    t["sweep_kernel_ms"] = st["sweep_kernel_ms"]
    t["precompute_ms"] = st["precompute_ms"]
    t0 = time.perf_counter(); done[0][2].write_corrected_fits([-1], os.path.join(d, "w.fits")); t["write_fits_ms"] = 1e3 * (time.perf_counter() - t0)
    from euispice_coreg_amd.hdrshift import AlignmentResults
    t0 = time.perf_counter(); AlignmentResults(done[0][2].corr, lags.arrays[0], lags.arrays[1], None, None, None, "arcsec"); t["gauss_fit_ms"] = 1e3 * (time.perf_counter() - t0)
    print(json.dumps({"images": len(done), "n": n, "lags_per_image": 10000, "wall_s": dt, "images_per_s": len(done) / dt,
                      "lag_points_per_s": 10000 * len(done) / dt, "active_points": st["n_active_points"],
                      "max_abs_shift_error_arcsec": err.max(axis=0).tolist(), "stages": t,
                      "devices": [list(d) for d in __import__("euispice_coreg_amd.jitter_correction.jitter_correction",
                                                              fromlist=["session_devices"]).session_devices(None, 1)]}))


if __name__ == "__main__":
    main()
