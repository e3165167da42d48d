"""
The Rice codec under AddressSanitizer + UndefinedBehaviorSanitizer on the host (tests/native/fuzz_rice.cpp).  The decoder
is one function for the host and for the GPU kernel's decoding lane (csrc/ricecomp.hpp), and the GPU pool has no
sanitizer: this is where "a corrupt stream is reported, never read past" is enforced byte by byte -- streams and
outputs allocated to the byte, truncations, bit flips, random bytes, tile tables pointing outside the heap, and the
round-trip property on random tiles of every pixel width.
"""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_codec_round_trips_and_survives_corrupt_streams_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_rice")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                         "-Wno-unknown-pragmas", os.path.join(HERE, "native", "fuzz_rice.cpp"), "-o", exe],
                        capture_output=True, text=True)
    if cc.returncode != 0 and "sanitize" in cc.stderr and "cannot find" in cc.stderr:
        pytest.skip("no sanitizer runtime for g++ here")
    assert cc.returncode == 0, cc.stderr[-2000:]
    for seed in (11, 12):
        r = subprocess.run([exe, "20000", str(seed)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok: 20000 iterations" in r.stdout, (r.stdout + r.stderr)[-3000:]
