#!/usr/bin/env python3
"""The floor of the PCIe-inclusive call ("pcie_inclusive <= 3.5 ms or a measured floor").

One call of `Alignment` on host images = upload of the image to align (16 MiB float32) + upload / preparation of the
reference (the rectangle the grid can touch) + precompute + sweep + finalize + correlation map on the host.  This script
measures, with calls issued back to back as bench.py's `pcie_inclusive` leg issues them (median and best of 12 after two
warm-ups), what each dependency costs and what a band-wise hand-over (sweep of the tile groups of band b starting when
bands <= b have landed) could remove AT MOST:

    whole           upload + reference preparation + sweep, host map out           (= pcie_inclusive)
    no_upload       the same call with the image to align already resident          (reference still re-prepared)
    resident        sweep only, host map out                                         (nothing re-prepared)
    upload          the upload alone, to completion
    first_band      the first quarter of the image alone (what must land before any sample can be taken)
    kernel          k_sweep inside each of those calls (HIP events of the library)

A perfect band-wise pipeline hides (upload - first_band) of `whole` and nothing else: floor = no_upload + first_band.
usage: python profiles/pcie_floor.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import _lib, synthetic  # noqa: E402

small, hs, large, hl, _ = synthetic.make_scene()
small32, large32 = small.astype(np.float32), large.astype(np.float32)
band = np.ascontiguousarray(small32[:512])
grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048), numpy_lat_trig=True)
lags = _lib.LagSet(np.arange(-30, 30, 1.0), np.arange(-30, 30, 1.0), None, None, None)
h = _lib.CoregHandle(0)
h2 = _lib.CoregHandle(0)   # (the band upload goes to a context of its own: the sweeps keep their image)


def timed(fn, n=12, kernel=False):
    ts, ks = [], []
    for k in range(n + 2):
        h.synchronize()
        t0 = time.perf_counter()
        fn()
        h.synchronize()
        dt = time.perf_counter() - t0
        if k >= 2:
            ts.append(1e3 * dt)
            if kernel:
                st = h.last_stats()
                ks.append((st["sweep_kernel_ms"], st["precompute_ms"]))
    out = {"best_ms": round(min(ts), 3), "median_ms": round(float(np.median(ts)), 3)}
    if ks:
        out["kernel_ms_median"] = round(float(np.median([k[0] for k in ks])), 3)
        out["precompute_ms_median"] = round(float(np.median([k[1] for k in ks])), 3)
    return out


def whole():
    h.set_small(small32)
    h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    return h.sweep_carrington(hs, grid, 1.004, lags)


def no_upload():
    h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    return h.sweep_carrington(hs, grid, 1.004, lags)


res = {}
res["whole"] = timed(whole, kernel=True)
want = whole()
res["no_upload"] = timed(no_upload, kernel=True)
assert np.array_equal(no_upload(), want, equal_nan=True)
res["resident"] = timed(lambda: h.sweep_carrington(hs, grid, 1.004, lags), kernel=True)
res["upload"] = timed(lambda: h.set_small(small32))
res["prepare_reference"] = timed(lambda: h.prepare_reference_carrington(large32, hl, grid, 1.004, 2))


def first_band():
    h2.set_small(band)
    h2.synchronize()


res["first_band"] = timed(first_band)
# steady state of the resident loop (device output, nothing read back: what bench.py's `value` times), for the kernel
import torch  # noqa: E402
out_dev = torch.empty(lags.size, dtype=torch.float64, device="cuda")
h.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(30):
    h.sweep_carrington(hs, grid, 1.004, lags, out_dev_ptr=out_dev.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    h.sweep_carrington(hs, grid, 1.004, lags, out_dev_ptr=out_dev.data_ptr())
torch.cuda.synchronize()
res["steady_state_ms_per_sweep"] = round(1e3 * (time.perf_counter() - t0) / 100, 3)
w, nu, fb, up = (res[k]["median_ms"] for k in ("whole", "no_upload", "first_band", "upload"))
res["floor_of_a_band_wise_hand_over_ms"] = round(nu + fb, 3)
res["what_it_could_remove_at_most_ms"] = round(w - (nu + fb), 3)
res["fixed_cost_of_a_call_beyond_its_kernel_ms"] = round(res["no_upload"]["median_ms"] - res["no_upload"]["kernel_ms_median"], 3)
for k, v in res.items():
    print(k, json.dumps(v), flush=True)
print("# target 3.5 ms:", "below the call with NO upload at all" if nu > 3.5 else "above the call without upload",
      f"(no_upload {nu} ms = kernel {res['no_upload']['kernel_ms_median']} + reference preparation / precompute / finalize / "
      f"read-back {res['fixed_cost_of_a_call_beyond_its_kernel_ms']}); a perfect band pipeline: {round(nu + fb, 3)} ms")
