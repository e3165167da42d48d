#!/usr/bin/env python3
"""Emulate the per-rank work of N-way lag-plane block sharding of the headline sweep on ONE GPU: time the slice of the
rank with the largest block (device-resident output, asynchronous calls, as bench.py does) with one and with two
sweeps in flight -> predicted T(N) without the all-gather."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from euispice_coreg_amd import _lib, synthetic, parallel
small, hs, large, hl, truth = synthetic.make_scene()
lag = np.arange(-30, 30, 1.0)
grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
streams = [torch.cuda.Stream() for _ in range(2)]
hh = []
for s in streams:
    h = _lib.CoregHandle(0)
    h.set_stream(s.cuda_stream)
    h.set_small(small)
    h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
    hh.append(h)
outs = [torch.empty(3600, dtype=torch.float64, device="cuda") for _ in range(2)]
t1 = {}
for world in (1, 2, 4, 8):
    lo1, hi1, lo2, hi2 = parallel.block_bounds(lag.size, lag.size, world, 0)
    sub = _lib.LagSet(lag[lo1:hi1], lag[lo2:hi2], None, None, None)
    for nstream in (1, 2):
        best = 1e9
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for rep in range(24):
                k = rep % nstream
                hh[k].sweep_carrington(hs, grid, 1.004, sub, out_dev_ptr=outs[k].data_ptr())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 24
            if it: best = min(best, dt)
        t1.setdefault(nstream, best)
        print(f"N={world} sweeps in flight={nstream}: {best*1e3:.3f} ms/step -> {3600/best:,.0f} lag-points/s, "
              f"efficiency {t1[nstream]/best/world:.2f}", flush=True)

# the same with the GRID sharded (every rank sweeps all 3600 lags over 1/N of the points): sweep + copy of the six sums
# + finalisation of (stand-in) reduced sums; the all-reduce of 6 x slots doubles itself is not emulated
full = _lib.LagSet(lag, lag, None, None, None)
sums = [None, None]
for world in (2, 4, 8):
    for h in hh:
        h.set_point_shard(0, world)
    for nstream in (1, 2):
        best = 1e9
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for rep in range(24):
                k = rep % nstream
                with torch.cuda.stream(streams[k]):
                    hh[k].sweep_carrington(hs, grid, 1.004, full, out_dev_ptr=outs[k].data_ptr())
                    if sums[k] is None:
                        sums[k] = torch.empty(hh[k].sums_size(), dtype=torch.float64, device="cuda")
                    hh[k].copy_sums(sums[k].data_ptr())
                    hh[k].finalize_sums(sums[k].data_ptr(), 3600, out_dev_ptr=outs[k].data_ptr())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 24
            if it: best = min(best, dt)
        print(f"N={world} POINT shares, sweeps in flight={nstream}: {best*1e3:.3f} ms/step -> {3600/best:,.0f} "
              f"lag-points/s, efficiency {t1[nstream]/best/world:.2f} (before the all-reduce of {sums[0].numel()} doubles)",
              flush=True)
for h in hh:
    h.set_point_shard(0, 1)
