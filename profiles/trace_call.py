import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from euispice_coreg_amd import _lib, synthetic
small, hs, large, hl, _ = synthetic.make_scene()
small32, large32 = small.astype(np.float32), large.astype(np.float32)
grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
lags = _lib.LagSet(np.arange(-30, 30, 1.0), np.arange(-30, 30, 1.0), None, None, None)
h = _lib.CoregHandle(0)
def whole(a):
    h.set_option("async_upload", a)
    h.set_small(small32)
    h.set_option("async_upload", 0)
    h.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    return h.sweep_carrington(hs, grid, 1.004, lags)
for a in (0, 1):
    for _ in range(6): whole(a)
    h.synchronize()
    print(f"==== async_upload={a}", file=sys.stderr, flush=True)
    os.environ["COREG_TRACE_MARK"] = "1"
    whole(a)
