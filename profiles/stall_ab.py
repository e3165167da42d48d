#!/usr/bin/env python3
"""Where the 2.5-3.0 s of round 5's stalled-collective test went: the recovery of
`coreg_multi` from a collective that never completes, with the release flag of the stand-in kernel (a) as round 5 had
it -- plain page-locked memory (hipHostMallocPortable), read by a volatile load -- and (b) as it is now -- COHERENT
page-locked memory read by a system-scope atomic load.  Phases from coreg_multi_rccl_status; the stand-in gives up by
itself after 12 s.  One GPU, one-rank RCCL group.   usage: python profiles/stall_ab.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["COREG_MULTI_FORCE_RCCL"] = "1"
from euispice_coreg_amd import _lib  # noqa: E402
from tests import helpers as H  # noqa: E402

small, hs, large, hl, _ = H.scene()
lags = _lib.LagSet(17.0 + np.arange(12) - 6.0, -9.0 + np.arange(12) - 6.0, None, None, None)
grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, (72, 64))
for legacy in (0, 1, 0):
    os.environ["COREG_RCCL_TEST_STALL_LEGACY"] = str(legacy)
    os.environ.pop("COREG_RCCL_TEST_STALL", None)
    with _lib.MultiHandle(device_ids=[0]) as m:
        if m.collective != "rccl":
            print("no RCCL runtime in this process")
            break
        m.set_small(small)
        m.prepare_reference_carrington(large, hl, grid, 1.004, 2)
        want = m.sweep_carrington(hs, grid, 1.004, lags)
        os.environ["COREG_RCCL_TEST_STALL"] = "1"
        os.environ["COREG_RCCL_WAIT_SECONDS"] = "0.4"
        t0 = time.perf_counter()
        got = m.sweep_carrington(hs, grid, 1.004, lags)
        dt = time.perf_counter() - t0
        print(f"legacy_flag={legacy}: call {dt:.3f} s, map equal {np.array_equal(got, want, equal_nan=True)}; {m.rccl_status}",
              flush=True)
