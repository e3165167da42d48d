#!/usr/bin/env python3
"""
Wall time of the drop-in call itself -- FITS files in, AlignmentResults out -- on the headline workload (2048^2 image to
align, 3072^2 reference, float32 FITS, 2048^2 Carrington grid, 60 x 60 CRVAL lags), next to what bench.py's
`pcie_inclusive` measures underneath it (host arrays in, map out).  Three calls: the first in a process (library load,
context, buffers), a second Alignment on the same files (warm context; the prepared reference is still resident and its
file is not decoded again), a third on a different image to align against the same reference.
usage: python profiles/api_timing.py          ->  one JSON line
"""
import cProfile
import io
import json
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import synthetic  # noqa: E402
from euispice_coreg_amd.hdrshift import Alignment  # noqa: E402
from euispice_coreg_amd.utils import fits_io  # noqa: E402


def main():
    d = tempfile.mkdtemp(prefix="coreg_api_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    small, hs, large, hl, truth = synthetic.make_scene()
    p_small, p_small2, p_large = (os.path.join(d, n) for n in ("hri.fits", "hri2.fits", "fsi.fits"))
    fits_io.write_images(p_small, [(None, {}), (small.astype(np.float32), hs)])
    fits_io.write_images(p_small2, [(None, {}), ((small * 1.01).astype(np.float32), hs)])
    fits_io.write_images(p_large, [(None, {}), (large.astype(np.float32), hl)])
    lag = np.arange(-30, 30, 1.0)

    def call(path, profile=False):
        t0 = time.perf_counter()
        A = Alignment(large_fov_known_pointing=p_large, small_fov_to_correct=path, lag_crval1=lag, lag_crval2=lag,
                      lag_cdelt1=[0], lag_cdelt2=[0], lag_crota=[0], parallelism=True)
        t1 = time.perf_counter()
        pr = cProfile.Profile() if profile else None
        if pr:
            pr.enable()
        res = A.align_using_carrington(lonlims=(200, 300), latlims=(-20, 20), shape=(2048, 2048))
        if pr:
            pr.disable()
        t2 = time.perf_counter()
        shift = res.shift_arcsec if hasattr(res, "shift_arcsec") else None
        top = None
        if pr:
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
            top = [ln.strip() for ln in s.getvalue().splitlines() if ln.strip() and ("{" in ln or ".py" in ln)][:14]
        return {"constructor_ms": 1e3 * (t1 - t0), "align_ms": 1e3 * (t2 - t1),
                "sweep_kernel_ms": A.last_stats["sweep_kernel_ms"], "shift": None if shift is None else list(map(float, np.ravel(shift)[:2])),
                "profile_top": top}

    out = {"workload": "headline through Alignment.align_using_carrington, float32 FITS files in /dev/shm",
           "first_call": call(p_small), "same_files_again": call(p_small),
           "other_image_same_reference": call(p_small2), "other_image_profiled": call(p_small, profile=True)}
    # best of a few warm calls
    warm = [call(p_small2 if i % 2 else p_small)["align_ms"] for i in range(6)]
    out["warm_calls_ms"] = warm
    # the same with headers as long as the instrument's (about 250 cards instead of the scene's 35), on files the process
    # has not seen: every call parses both headers once (cached per file state afterwards)
    fat = {f"KEYF{k:03d}": 1.2345e-3 * k for k in range(120)}
    fat.update({f"KEYS{k:03d}": f"string value number {k} / with a slash" for k in range(70)})
    fat.update({f"KEYI{k:03d}": 1000 * k for k in range(25)})
    p_l250 = os.path.join(d, "fsi_250.fits")
    fits_io.write_images(p_l250, [(None, {}), (large.astype(np.float32), dict(hl, **fat))])
    call_fat = []
    for i in range(4):
        p_s250 = os.path.join(d, f"hri_250_{i}.fits")
        fits_io.write_images(p_s250, [(None, {}), ((small * (1 + 0.01 * i)).astype(np.float32), dict(hs, **fat))])
        call_fat.append(call(p_s250)["align_ms"])
        os.remove(p_s250)
    out["warm_calls_ms_250_card_headers_new_image_each"] = call_fat
    call_new = []  # (the control: a new image each call, the scene's short headers)
    for i in range(4):
        p_s35 = os.path.join(d, f"hri_35_{i}.fits")
        fits_io.write_images(p_s35, [(None, {}), ((small * (1 + 0.01 * i)).astype(np.float32), hs)])
        call_new.append(call(p_s35)["align_ms"])
        os.remove(p_s35)
    out["warm_calls_ms_new_image_each"] = call_new
    # the two host stages of a warm call on their own (no profiler: cProfile inflates the many small scipy calls)
    from euispice_coreg_amd.hdrshift.alignment_results import AlignmentResults
    A = Alignment(large_fov_known_pointing=p_large, small_fov_to_correct=p_small, lag_crval1=lag, lag_crval2=lag,
                  lag_cdelt1=[0], lag_cdelt2=[0], lag_crota=[0], parallelism=True)
    corr = A.align_using_carrington(lonlims=(200, 300), latlims=(-20, 20), shape=(2048, 2048), return_type="corr")
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        AlignmentResults(corr, lag, lag, [0], [0], [0], "arcsec")
        t.append(1e3 * (time.perf_counter() - t0))
    out["gaussian_fit_ms"] = min(t)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        fits_io.read_image(p_small, -1)
        t.append(1e3 * (time.perf_counter() - t0))
    out["fits_decode_small_host_ms"] = min(t)  # what a host decode costs (no longer on the call's path)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        raw = fits_io.open_raw(p_small, -1)
        t.append(1e3 * (time.perf_counter() - t0))
        raw.close()
    out["fits_decode_small_ms"] = min(t)       # header parse + memory map: all the host does with the pixels now
    from euispice_coreg_amd import _lib
    h = _lib.shared_handle(-1, 0)
    raw = fits_io.open_raw(p_small, -1)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        h.set_small(raw)
        h.synchronize()
        t.append(1e3 * (time.perf_counter() - t0))
    out["raw_upload_and_gpu_decode_ms"] = min(t)
    res = AlignmentResults(corr, lag, lag, [0], [0], [0], "arcsec", image_to_align_path=p_small)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        res.write_corrected_fits([-1], os.path.join(d, "out.fits"))
        t.append(1e3 * (time.perf_counter() - t0))
    out["write_corrected_fits_ms"] = min(t)
    os.remove(os.path.join(d, "out.fits"))
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        AlignmentResults(corr, lag, lag, [0], [0], [0], "arcsec", fit="scipy")
        t.append(1e3 * (time.perf_counter() - t0))
    out["gaussian_fit_scipy_ms"] = min(t)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        A.align_using_carrington(lonlims=(200, 300), latlims=(-20, 20), shape=(2048, 2048), return_type="corr")
        t.append(1e3 * (time.perf_counter() - t0))
    out["align_without_fit_ms"] = min(t)
    print(json.dumps(out))
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
