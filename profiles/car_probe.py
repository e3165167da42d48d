import sys, os, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
from euispice_coreg_amd import _lib
from tests import test_gpu_car as T
h = _lib.CoregHandle(-1)
for seed in [int(v) for v in sys.argv[1:]]:
    got, want, lags, order, hs = T._sub_map_case(h, seed)
    d = np.abs(got - want)
    print("seed", seed, "order", order, "crota", hs["CROTA"], "cdelt", hs["CDELT1"], hs["CDELT2"], "crval", hs["CRVAL1"], hs["CRVAL2"], "shape", hs["NAXIS2"], hs["NAXIS1"], "lonpole", hs.get("LONPOLE"), "tap", h.last_tap_fix())
    print("  l1", lags[0].tolist(), "l2", lags[1].tolist(), "crota lags", lags[4])
    for b in np.argwhere(~(d <= 1e-7)): print("   bad", tuple(int(v) for v in b[:5]), "lag", lags[0][b[0]], lags[1][b[1]], lags[4][b[4]], got[tuple(b)], want[tuple(b)])
