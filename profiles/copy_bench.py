#!/usr/bin/env python3
"""How fast can the 16 MiB data unit of a corrected FITS file be copied on this box?  (write_corrected_fits copies data
units as they are: utils/fits_io.py rewrite_with_corrected_headers.)  One in-kernel copy against ranges copied by
several threads into a pre-sized file, on /dev/shm and on the temporary directory.   -> one JSON line"""
import json
import os
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def best(fn, n=10):
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return 1e3 * min(t)


def main():
    out = {}
    blob = np.random.default_rng(0).integers(0, 255, 16 << 20, dtype=np.uint8).tobytes()
    pool = ThreadPoolExecutor(8)
    for name, d in (("shm", "/dev/shm"), ("tmp", tempfile.gettempdir())):
        if not os.path.isdir(d):
            continue
        src, dst = os.path.join(d, "coreg_copy_src.bin"), os.path.join(d, "coreg_copy_dst.bin")
        open(src, "wb").write(b" " * 8640 + blob)
        n = os.path.getsize(src)

        def one():
            if os.path.exists(dst):
                os.remove(dst)
            with open(src, "rb") as fi, open(dst, "wb") as fo:
                done = 0
                while done < n:
                    done += os.copy_file_range(fi.fileno(), fo.fileno(), n - done, done, done)

        def many(nt):
            def run():
                if os.path.exists(dst):
                    os.remove(dst)
                fi = os.open(src, os.O_RDONLY)
                fo = os.open(dst, os.O_WRONLY | os.O_CREAT, 0o644)
                os.ftruncate(fo, n)
                chunk = (n + nt - 1) // nt

                def part(k):
                    off, end = k * chunk, min(n, (k + 1) * chunk)
                    while off < end:
                        off += os.copy_file_range(fi, fo, end - off, off, off)
                list(pool.map(part, range(nt)))
                os.close(fi)
                os.close(fo)
            return run
        res = {"one_copy_file_range_ms": best(one)}
        for nt in (2, 4, 8):
            res[f"{nt}_threads_ms"] = best(many(nt))
        out[name] = res
        os.remove(src)
        os.remove(dst)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
