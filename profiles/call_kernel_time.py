#!/usr/bin/env python3
"""Why is the sweep kernel slower inside a whole call (3.5-3.8 ms) than in the resident loop of bench.py (3.13 ms)?
Prints the kernel time (HIP events around k_sweep) of: resident sweeps back to back; whole calls back to back (upload +
reference preparation + sweep); whole calls with only the upload; with only the preparation; resident sweeps with a host
pause between them."""
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
h.set_small(small32)
h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)


def run(name, fn, n=12):
    ks, ws = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ws.append(1e3 * (time.perf_counter() - t0))
        ks.append(h.last_stats()["sweep_kernel_ms"])
    print(f"{name:46s} kernel ms min {min(ks):.3f} median {np.median(ks):.3f} last {ks[-1]:.3f} | call ms median {np.median(ws):.3f}",
          flush=True)


def sweep():
    return h.sweep_carrington(hs, grid, 1.004, lags)


def whole():
    h.set_small(small32)
    h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    return sweep()


def upload_only():
    h.set_small(small32)
    return sweep()


def prepare_only():
    h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    return sweep()


def paused(ms):
    def f():
        time.sleep(ms * 1e-3)
        return sweep()
    return f


run("resident sweeps, back to back", sweep)
run("whole calls, back to back", whole)
run("upload + sweep", upload_only)
run("prepare + sweep", prepare_only)
for ms in (0.5, 2, 10, 50):
    run(f"resident sweeps, {ms} ms host pause before each", paused(ms))
h.set_option("overlap_upload", 0)
run("whole calls, upload on the handle's stream", whole)
