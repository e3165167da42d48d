"""
The library's Gaussian fit (csrc/fit.hpp) under AddressSanitizer + UndefinedBehaviorSanitizer (tests/native/fuzz_fit.cpp):
random peaks and the degenerate maps real sweeps produce (flat, one point, collinear points, NaN / Inf samples, starts on a
bound, bounds a hair apart, infinite bounds) -- no sanitizer report, scipy's status codes, results inside the bounds,
bit-reproducible.
"""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_fit_survives_degenerate_maps_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_fit")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                         "-Wno-unknown-pragmas", os.path.join(HERE, "native", "fuzz_fit.cpp"), "-o", exe],
                        capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "cannot find" in cc.stderr:
        pytest.skip("no sanitizer runtime for g++ here")
    assert cc.returncode == 0, cc.stderr[-2000:]
    r = subprocess.run([exe, "3000", "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok: 3000 iterations" in r.stdout, (r.stdout + r.stderr)[-3000:]
