#!/usr/bin/env python3
"""
Where the FIRST drop-in call of a process spends its time (the user who runs one alignment from a script pays this
once): interpreter imports, dlopen of the library, HIP runtime + context, first allocations and pinned staging, code-object
load of the first kernels, and the call itself.  Every phase is timed in a fresh process; the warm call follows.
usage: python profiles/first_call.py        -> one JSON line
"""
import json
import os
import sys
import tempfile
import time

T0 = time.perf_counter()
import numpy as np  # noqa: E402

T_NUMPY = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import _lib, synthetic  # noqa: E402
from euispice_coreg_amd.hdrshift import Alignment  # noqa: E402
from euispice_coreg_amd.utils import fits_io  # noqa: E402

T_PKG = time.perf_counter()


def main():
    out = {"import_numpy_ms": 1e3 * (T_NUMPY - T0), "import_package_ms": 1e3 * (T_PKG - T_NUMPY)}
    d = tempfile.mkdtemp(prefix="coreg_first_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    small, hs, large, hl, _ = synthetic.make_scene()
    p_small, p_large = os.path.join(d, "hri.fits"), os.path.join(d, "fsi.fits")
    fits_io.write_images(p_small, [(None, {}), (small.astype(np.float32), hs)])
    fits_io.write_images(p_large, [(None, {}), (large.astype(np.float32), hl)])
    lag = np.arange(-30, 30, 1.0)

    def lap(name, fn):
        t = time.perf_counter()
        r = fn()
        out[name] = 1e3 * (time.perf_counter() - t)
        return r

    if "--phases" in sys.argv:
        # the same work through the binding, phase by phase
        lap("dlopen_library_ms", _lib.load_library)
        h = lap("create_handle_ms", lambda: _lib.CoregHandle(0))
        rs, rl = fits_io.open_raw(p_small, -1), fits_io.open_raw(p_large, -1)
        lap("set_small_first_ms", lambda: h.set_small(rs))
        lap("set_small_again_ms", lambda: h.set_small(rs))
        grid = _lib.Grid((200, 300), (-20, 20), (2048, 2048))
        lap("prepare_reference_first_ms", lambda: h.prepare_reference_carrington(rl, hl, grid, 1.004, 2))
        lap("prepare_reference_again_ms", lambda: h.prepare_reference_carrington(rl, hl, grid, 1.004, 2))
        lags = _lib.LagSet(lag, lag, None, None, None)
        lap("sweep_first_ms", lambda: h.sweep_carrington(hs, grid, 1.004, lags))
        lap("sweep_again_ms", lambda: h.sweep_carrington(hs, grid, 1.004, lags))
        h.close()
    else:
        def call():
            A = Alignment(large_fov_known_pointing=p_large, small_fov_to_correct=p_small, lag_crval1=lag, lag_crval2=lag,
                          lag_cdelt1=[0], lag_cdelt2=[0], lag_crota=[0], parallelism=True)
            return A.align_using_carrington(lonlims=(200, 300), latlims=(-20, 20), shape=(2048, 2048))
        lap("first_call_ms", call)
        lap("second_call_ms", call)
        lap("third_call_ms", call)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
