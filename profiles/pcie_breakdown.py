#!/usr/bin/env python3
"""Where the PCIe-inclusive step goes: upload of the image to align, upload + preparation of the reference, sweep with
the map copied back -- each timed to completion (stream synchronised), float32 inputs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import _lib, synthetic  # noqa: E402

small, hs, large, hl, _ = synthetic.make_scene()
small32, large32 = small.astype(np.float32), large.astype(np.float32)
grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
lags = _lib.LagSet(np.arange(-30, 30, 1.0), np.arange(-30, 30, 1.0), None, None, None)
h = _lib.CoregHandle(0)
if os.environ.get("COREG_OVERLAP_UPLOAD") == "0":
    h.set_option("overlap_upload", 0)


def t(fn, n=6):
    best = 1e9
    for _ in range(n):
        h.synchronize()
        t0 = time.perf_counter()
        fn()
        h.synchronize()
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best


print("set_small f32 (16 MiB)           %.3f ms" % t(lambda: h.set_small(small32)))
print("prepare_reference f32 (36 MiB)   %.3f ms" % t(lambda: h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)))
print("set_small f64 (32 MiB)           %.3f ms" % t(lambda: h.set_small(small)))
h.set_small(small32)
print("sweep, host map out              %.3f ms" % t(lambda: h.sweep_carrington(hs, grid, 1.004, lags)))
def whole():
    h.set_small(small32)
    h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    return h.sweep_carrington(hs, grid, 1.004, lags)
print("whole call                       %.3f ms" % t(whole, 12))
def whole_async():
    h.set_option("async_upload", 1)
    h.set_small(small32)
    h.set_option("async_upload", 0)
    h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    return h.sweep_carrington(hs, grid, 1.004, lags)
print("whole call, async upload         %.3f ms" % t(whole_async, 12))
want = whole()
assert np.array_equal(whole_async(), want, equal_nan=True)
def whole_ref_first():
    h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    h.set_small(small32)
    return h.sweep_carrington(hs, grid, 1.004, lags)
print("whole call, reference first      %.3f ms" % t(whole_ref_first, 12))
import ctypes
buf = np.empty(36 << 20, dtype=np.uint8); src = np.random.default_rng(0).integers(0, 255, 36 << 20, dtype=np.uint8)
t0 = time.perf_counter(); np.copyto(buf, src); print("numpy memcpy 36 MiB 1 thread     %.3f ms" % (1e3 * (time.perf_counter() - t0)))
whole_async()
st = h.last_stats()
print("inside the async whole call: sweep_kernel_ms %.3f precompute_ms %.3f total_gpu_ms %.3f" % (
    st["sweep_kernel_ms"], st.get("precompute_ms", float("nan")), st["total_gpu_ms"]))
