#!/usr/bin/env python3
"""What a file the process has never mapped costs on its way to the GPU: open (header parse + mmap) and upload of a
2048^2 float32 image, for a file that was mapped before and for files written a moment ago, with the mapping's page
tables filled at mmap time (MAP_POPULATE, COREG_MMAP_POPULATE=1) or by the faults of the copy threads (the default).
usage: python profiles/fresh_file_upload.py   -> one JSON line"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import _lib, synthetic  # noqa: E402
from euispice_coreg_amd.utils import fits_io  # noqa: E402


def main():
    d = tempfile.mkdtemp(prefix="coreg_fresh_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    small, hs, _, _, _ = synthetic.make_scene()
    img = small.astype(np.float32)
    h = _lib.CoregHandle(0)
    out = {}

    def one(p):
        t0 = time.perf_counter()
        raw = fits_io.open_raw(p, -1)
        t1 = time.perf_counter()
        h.set_small(raw)
        h.synchronize()
        t2 = time.perf_counter()
        raw.close()
        return 1e3 * (t1 - t0), 1e3 * (t2 - t1)

    for populate in ("1", "0"):
        os.environ["COREG_MMAP_POPULATE"] = populate
        p0 = os.path.join(d, "old.fits")
        fits_io.write_images(p0, [(None, {}), (img, hs)])
        one(p0)
        old = [one(p0) for _ in range(5)]
        fresh = []
        for i in range(5):
            p = os.path.join(d, f"new{i}.fits")
            fits_io.write_images(p, [(None, {}), (img * (1 + 0.01 * i), hs)])
            fresh.append(one(p))
            os.remove(p)
        key = "populate" if populate == "1" else "no_populate"
        out[key] = {"mapped_before_open_ms": min(a for a, _ in old), "mapped_before_upload_ms": min(b for _, b in old),
                    "fresh_open_ms": min(a for a, _ in fresh), "fresh_upload_ms": min(b for _, b in fresh),
                    "fresh_total_ms": min(a + b for a, b in fresh), "mapped_before_total_ms": min(a + b for a, b in old)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
