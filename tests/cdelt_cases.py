"""The intended CDELT-lag semantics as the REFERENCE's own code gives them (tests/golden/cdelt_intended_golden.{npz,json},
cdelt_intended_cfg5.npz; generator tests/golden/make_golden_cdelt_intended.py: `Alignment` at its zero lag, or sweeping
its working CRVAL / CROTA lags, on files whose header went through `AlignCommonUtil.correct_pointing_header`,
Util.py:161-215).  Shared by the CPU test (oracle, host ABI) and the GPU test (HIP path)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CARDS = ("CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "CROTA", "PC1_1", "PC1_2", "PC2_1", "PC2_2")
TOL = {"carrington": 1e-10, "helio_serial": 1e-7, "helio_parallel": 1e-7}

_cache = {}


def load():
    if not _cache:
        _cache["npz"] = np.load(os.path.join(GOLDEN, "cdelt_intended_golden.npz"))
        with open(os.path.join(GOLDEN, "cdelt_intended_golden.json")) as f:
            _cache["meta"] = json.load(f)
    return _cache["npz"], _cache["meta"]


def scene_names():
    return sorted(load()[1]["scenes"])


def scene(name):
    """-> small, hdr_small, large, hdr_large (float32 pixels and cards as the reference read them), sub-map, meta"""
    g, m = load()
    sc = m["scenes"][name]
    return (g[f"scene/{name}/small"], dict(sc["hdr_small"]), g[f"scene/{name}/large"], dict(sc["hdr_large"]),
            g[f"scene/{name}/submap"], sc)


def entries(name, frame):
    """-> index [n, 5] into the scene's 5-D lag grid (crval1, crval2, cdelt1, cdelt2, crota), reference coefficients [n],
    mode [n] ('zero_lag' | 'swept')"""
    _, m = load()
    es = [e for e in m["entries"] if e["scene"] == name and e["frame"] == frame]
    return (np.array([e["index"] for e in es], dtype=np.int64).reshape(-1, 5), np.array([e["corr"] for e in es]),
            [e["mode"] for e in es])


def frames(name):
    _, m = load()
    return sorted({e["frame"] for e in m["entries"] if e["scene"] == name})


def corrected(name):
    _, m = load()
    return [c for c in m["corrected"] if c["scene"] == name]


def lag_axes(sc):
    """The scene's five lag axes in the order of the sweep's index (crval1, crval2, cdelt1, cdelt2, crota), in arcsec /
    degrees of rotation as handed to `Alignment`."""
    return [np.asarray(a, dtype=np.float64) for a in sc["axes"]]
