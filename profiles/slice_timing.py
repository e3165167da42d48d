#!/usr/bin/env python3
"""Emulate the per-rank work of N-way lag-plane block sharding of the headline sweep on ONE GPU: time rank r's slice for every r
(device-resident output, as bench.py does) and report max over ranks -> predicted T(N) without the all-gather."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from euispice_coreg_amd import _lib, synthetic

small, hs, large, hl, truth = synthetic.make_scene()
lag = np.arange(-30, 30, 1.0)
lags = _lib.LagSet(lag, lag, None, None, None)
grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
h = _lib.CoregHandle(0)
h.set_stream(torch.cuda.current_stream().cuda_stream)
h.set_small(small)
h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
L = lags.size
out = torch.empty(L, dtype=torch.float64, device="cuda")
t1 = None
for world in (1, 2, 4, 8):
    from euispice_coreg_amd import parallel
    worst = 0.0
    for r in range(world):
        lo1, hi1, lo2, hi2 = parallel.block_bounds(lag.size, lag.size, world, r)
        sub = _lib.LagSet(lag[lo1:hi1], lag[lo2:hi2], None, None, None)
        best = 1e9
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for rep in range(5):  # back-to-back asynchronous calls, as in bench.py
                h.sweep_carrington(hs, grid, 1.004, sub, out_dev_ptr=out.data_ptr())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            if it: best = min(best, dt)
        st = h.last_stats()
        worst = max(worst, best)
    if world == 1: t1 = worst
    print(f"N={world}: per-rank wall max {worst*1e3:.3f} ms (kernel {st['sweep_kernel_ms']:.3f}, gpu {st['total_gpu_ms']:.3f})  "
          f"-> {L/worst:,.0f} lag-points/s, efficiency {t1/worst/world:.2f}", flush=True)
h.close()
