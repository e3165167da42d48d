#!/usr/bin/env python3
"""Soak: many drop-in calls in one process (both frames, alternating images, changing lag sets and grid shapes) with the
host RSS and the free device memory sampled along the way -- buffers are meant to be re-used, not to grow.
usage: python profiles/soak.py [n_calls]   -> one JSON line"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import psutil  # noqa: E402
import torch  # noqa: E402
from euispice_coreg_amd import synthetic  # noqa: E402
from euispice_coreg_amd.hdrshift import Alignment  # noqa: E402
from euispice_coreg_amd.utils import fits_io  # noqa: E402


def main():
    n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    d = tempfile.mkdtemp(prefix="coreg_soak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    small, hs, large, hl, truth = synthetic.make_scene(small_n=1024, large_n=1536)
    paths = []
    for k in range(3):
        p = os.path.join(d, f"hri{k}.fits")
        fits_io.write_images(p, [(None, {}), ((small * (1.0 + 0.01 * k)).astype(np.float32), hs)])
        paths.append(p)
    p_large = os.path.join(d, "fsi.fits")
    fits_io.write_images(p_large, [(None, {}), (large.astype(np.float32), hl)])
    proc = psutil.Process()
    rng = np.random.default_rng(0)
    samples, shifts = [], []
    t0 = time.perf_counter()
    for i in range(n_calls):
        n1, n2 = int(rng.integers(20, 61)), int(rng.integers(20, 61))
        lag1 = np.arange(n1) - n1 // 2 + 10.0
        lag2 = np.arange(n2) - n2 // 2 - 5.0
        crota = [0.0] if i % 3 else [0.0, 0.3]
        A = Alignment(large_fov_known_pointing=p_large, small_fov_to_correct=paths[i % 3], lag_crval1=lag1, lag_crval2=lag2,
                      lag_cdelt1=[0], lag_cdelt2=[0], lag_crota=crota, parallelism=True)
        if i % 2:
            res = A.align_using_helioprojective()
        else:
            g = int(rng.choice([512, 768, 1024]))
            res = A.align_using_carrington(lonlims=(228, 262), latlims=(-12, 22), shape=(g, g))
        shifts.append([float(v) for v in np.ravel(res.shift_arcsec)[:2]])
        if i % 25 == 0 or i == n_calls - 1:
            torch.cuda.synchronize()
            free, total = torch.cuda.mem_get_info()
            samples.append({"call": i, "rss_mib": proc.memory_info().rss / 2**20, "dev_used_mib": (total - free) / 2**20,
                            "s": time.perf_counter() - t0})
            print(f"[soak] call {i}: rss {samples[-1]['rss_mib']:.0f} MiB, device used {samples[-1]['dev_used_mib']:.0f} MiB",
                  file=sys.stderr, flush=True)
    sh = np.array(shifts)
    out = {"calls": n_calls, "wall_s": time.perf_counter() - t0, "samples": samples,
           "rss_growth_mib_after_first_50": samples[-1]["rss_mib"] - [s for s in samples if s["call"] >= 50][0]["rss_mib"],
           "dev_growth_mib_after_first_50": samples[-1]["dev_used_mib"] - [s for s in samples if s["call"] >= 50][0]["dev_used_mib"],
           "shift_spread_arcsec": [float(sh[:, 0].std()), float(sh[:, 1].std())]}
    print(json.dumps(out))
    for p in paths + [p_large]:
        os.remove(p)
    os.rmdir(d)


if __name__ == "__main__":
    main()
