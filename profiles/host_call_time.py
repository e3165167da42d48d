#!/usr/bin/env python3
"""Host side of an asynchronous sweep call (device output): how long the call itself takes with the GPU busy / idle, for the
whole headline lag set and for one rank's block at N = 8 -- is the host the limit when a sweep lasts 0.5 ms?
usage: python profiles/host_call_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from euispice_coreg_amd import _lib, synthetic, parallel
small, hs, large, hl, truth = synthetic.make_scene()
lag = np.arange(-30, 30, 1.0)
grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
h = _lib.CoregHandle(0)
s = torch.cuda.Stream(); h.set_stream(s.cuda_stream)
h.set_small(small); h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
out = torch.empty(3600, dtype=torch.float64, device="cuda")
for world in (1, 8):
    lo1, hi1, lo2, hi2 = parallel.block_bounds(lag.size, lag.size, world, 0)
    sub = _lib.LagSet(lag[lo1:hi1], lag[lo2:hi2], None, None, None)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); calls = []
        for rep in range(50):
            c0 = time.perf_counter()
            h.sweep_carrington(hs, grid, 1.004, sub, out_dev_ptr=out.data_ptr())
            calls.append(time.perf_counter() - c0)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N={world}: host call median {1e3*np.median(calls):.3f} ms, min {1e3*min(calls):.3f}; all 50 issued in {1e3*t_issue:.1f} ms; done in {1e3*dt:.1f} ms ({1e3*dt/50:.3f} ms/step)")
# host time alone with the GPU idle in between (sync after each)
for world in (8,):
    lo1, hi1, lo2, hi2 = parallel.block_bounds(lag.size, lag.size, world, 0)
    sub = _lib.LagSet(lag[lo1:hi1], lag[lo2:hi2], None, None, None)
    calls = []
    for rep in range(30):
        torch.cuda.synchronize(); c0 = time.perf_counter()
        h.sweep_carrington(hs, grid, 1.004, sub, out_dev_ptr=out.data_ptr())
        calls.append(time.perf_counter() - c0)
    print(f"N=8 block, GPU idle at call time: host call median {1e3*np.median(calls):.3f} ms")
