#!/bin/bash
# jitter session throughput, the end-to-end example (also with two virtual devices), group-count check for N = 8 blocks
O=$1; mkdir -p $O
timeout -k 10 500 python3 profiles/jitter_bench.py > $O/jitter_bench.json 2> $O/jitter.err || { tail $O/jitter.err; exit 1; }
cat $O/jitter_bench.json | cut -c1-900; echo
timeout -k 10 300 python3 examples/align_synthetic.py > $O/example.log 2>&1 || { tail $O/example.log; exit 1; }
tail -4 $O/example.log
COREG_VIRTUAL_DEVICES=2 timeout -k 10 300 python3 examples/align_synthetic.py > $O/example_v2.log 2>&1 || { tail $O/example_v2.log; exit 1; }
tail -4 $O/example_v2.log
TUNE_NLAG=60 timeout -k 10 300 python3 - > $O/groups_n8.log 2>&1 <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from euispice_coreg_amd import _lib, synthetic, parallel
small, hs, large, hl, truth = synthetic.make_scene()
lag = np.arange(-30, 30, 1.0)
grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
h = _lib.CoregHandle(0); h.set_stream(torch.cuda.current_stream().cuda_stream)
h.set_small(small); h.prepare_reference_carrington(large, hl, grid, 1.004, 2)
out = torch.empty(3600, dtype=torch.float64, device="cuda")
for world in (8, 4):
    lo1, hi1, lo2, hi2 = parallel.block_bounds(60, 60, world, 0)
    sub = _lib.LagSet(lag[lo1:hi1], lag[lo2:hi2], None, None, None)
    for ng in (0, 256, 384, 512, 768, 0):
        h.set_option("n_groups", ng)
        best = 1e9
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for rep in range(24):
                h.sweep_carrington(hs, grid, 1.004, sub, out_dev_ptr=out.data_ptr())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 24
            if it: best = min(best, dt)
        print(f"N={world} n_groups={ng}: {best*1e3:.3f} ms/step, kernel {h.last_stats()['sweep_kernel_ms']:.3f}", flush=True)
PY
cat $O/groups_n8.log
