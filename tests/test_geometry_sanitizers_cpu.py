"""
The host-side header arithmetic of the C ABI (csrc/geometry.hpp) under AddressSanitizer + UndefinedBehaviorSanitizer
(tests/native/fuzz_geometry.cpp): sane random headers and hostile ones (zero / denormal CDELT, singular PC, CRVAL at the
poles, NaN / Inf cards, huge multiples of 90 degrees) through shift_header, the homography and its lag family, the
restated wcslib TAN and CAR chains, the plate-carree maps and the Carrington tables -- no sanitizer report, identity maps
return the pixel they were given, homography == wcslib chain on sane headers, bit-reproducible.
"""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_header_arithmetic_survives_hostile_headers_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_geometry")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined,float-cast-overflow",
                         "-fno-sanitize-recover=undefined,float-cast-overflow", "-Wno-unknown-pragmas",
                         os.path.join(HERE, "native", "fuzz_geometry.cpp"), "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "cannot find" in cc.stderr:
        pytest.skip("no sanitizer runtime for g++ here")
    assert cc.returncode == 0, cc.stderr[-2000:]
    r = subprocess.run([exe, "20000", "7"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok: 20000 iterations" in r.stdout, (r.stdout + r.stderr)[-3000:]
