import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rectify_golden():
    return np.load(os.path.join(GOLDEN, "rectify_golden.npz"))


@pytest.fixture(scope="session")
def wcs_golden():
    return np.load(os.path.join(GOLDEN, "wcs_golden.npz"))


def golden_header(g, prefix):
    h = dict(zip([str(k) for k in g[prefix + "/keys"]], [float(v) for v in g[prefix + "/vals"]]))
    h["CUNIT1"] = h["CUNIT2"] = str(g[prefix + "/unit"])
    h["NAXIS1"], h["NAXIS2"] = int(h["NAXIS1"]), int(h["NAXIS2"])
    return h


def rectify_case(g, c):
    hdr = dict(zip([str(k) for k in g[c + "/hdr_keys"]], [float(v) for v in g[c + "/hdr_vals"]]))
    hdr["CUNIT1"] = hdr["CUNIT2"] = "arcsec"
    shape = [int(v) for v in g[c + "/shape"]]
    return dict(hdr=hdr, shape=shape, lonlims=[float(v) for v in g[c + "/lonlims"]],
                latlims=[float(v) for v in g[c + "/latlims"]], solar_r=float(g[c + "/solar_r"]),
                order=int(g[c + "/order"]), image=g[c + "/image"], nx=g[c + "/nx"], ny=g[c + "/ny"],
                resampled=g[c + "/resampled"])


@pytest.fixture(scope="session")
def gpu_handle():
    """One libcoreg_hip handle for the session; fails loudly (never falls back) if unavailable."""
    from euispice_coreg_amd import _lib
    h = _lib.CoregHandle(0)
    yield h
    h.close()


@pytest.fixture(scope="session")
def car_golden():
    return np.load(os.path.join(GOLDEN, "car_golden.npz"))


def car_header(g, name):
    """CAR header dict of a case of car_golden.npz."""
    h = dict(zip([str(k) for k in g[name + "/keys"]], [float(v) for v in g[name + "/vals"]]))
    for k in ("NAXIS1", "NAXIS2"):
        if k in h:
            h[k] = int(h[k])
    h.update(CTYPE1="CRLN-CAR", CTYPE2="CRLT-CAR", CUNIT1="deg", CUNIT2="deg")
    return h
