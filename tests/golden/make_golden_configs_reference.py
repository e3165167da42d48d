#!/opt/conda/bin/python3.9
"""
BASELINE.json configs[1] .. configs[4] AT THEIR STATED SIZES AS THE REFERENCE ITSELF RUNS THEM, each on a sub-lattice of
its lags (lag-points are independent of one another, so these are entries of the full maps) -- the reference's own
`Alignment(...).align_using_helioprojective / align_using_carrington(return_type='corr')` (`hdrshift/alignment.py`)
executed in the build container (side interpreter + load-time shims: `_reference_loader.py`):

    cfg2   2048^2 against 3072^2, helioprojective, parallelism=True (sub-map), lags [-30, 30]: 7 x 7 lattice (zero lag
           included) + 3 x 3 around the injected shift
    cfg3   the same pair on a 2048^2 Carrington grid, lags [-60, 60]: 7 x 7 lattice + 3 x 3 around the shift
    cfg4   SPICE-like raster 192 x 832 (header in degrees, lags in arcsec) against 3072^2, helioprojective,
           lags [-30, 30]^2 x CROTA [-1, 1]: 5 x 5 x 5 lattice + the injected shift
    cfg5   4096^2 Carrington grid, lags [-20, 20]^2 x CROTA [-0.5, 0.5], the d_cdelt1 = d_cdelt2 = 0 plane of the 5-D
           sweep (a d_cdelt2 != 0 lag-point raises in the reference, quirk Q2): 3 x 3 x 3

Three steps, so that the pixels are exactly those the tests regenerate (numpy 2.2 of the main interpreter):
    python tests/golden/make_golden_headline.py --dump-scene /tmp/headline_scene.npz
    python tests/golden/make_golden_configs_reference.py --dump-cfg4-scene /tmp/cfg4_scene.npz
    /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden_configs_reference.py /tmp/headline_scene.npz /tmp/cfg4_scene.npz
-> tests/golden/configs_reference.npz (per config: lag indices into the full map, the reference's coefficients; the
scene fingerprints; the header cards AS ASTROPY READ THEM BACK from the FITS files -- 4.3.1 writes floats with 16
significant digits, which moves the cfg4 header's degrees by an ulp here and there: enough to change which border pixels
wcslib's rounding noise drops at the zero lag, so the tests use these cards).  A few minutes on 8 cores.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def cfg4_fingerprint(np, small, large):
    return np.array([np.nansum(small), np.nansum(large), float(np.isnan(small).sum()), small[400, 100], large[1500, 1500]])


def dump_cfg4_scene(path):
    import numpy as np
    sys.path.insert(0, ROOT)
    from euispice_coreg_amd import synthetic
    small, hs, large, hl, truth = synthetic.make_scene(small_shape=(832, 192), small_cdelt=(4.0, 1.098), small_unit="deg",
                                                       large_n=3072)
    s32, l32 = small.astype(np.float32), large.astype(np.float32)
    assert np.array_equal(s32.astype(np.float64), small, equal_nan=True) and np.array_equal(l32.astype(np.float64), large)
    np.savez(path, small=s32, large=l32, hdr_small=np.array(json.dumps(hs)), hdr_large=np.array(json.dumps(hl)),
             fingerprint=cfg4_fingerprint(np, small, large))
    print("wrote", path)


def main():
    import tempfile
    import time
    import warnings
    sys.path.insert(0, HERE)
    import _reference_loader
    _reference_loader.load_reference()
    warnings.filterwarnings("ignore")
    import numpy as np
    from astropy.io import fits
    from euispice_coreg.hdrshift.alignment import Alignment

    tmp = tempfile.mkdtemp(prefix="golden_configs_")

    def write(tag, sc):
        hs, hl = json.loads(str(sc["hdr_small"])), json.loads(str(sc["hdr_large"]))
        ps, pl = os.path.join(tmp, tag + "_small.fits"), os.path.join(tmp, tag + "_large.fits")
        for p, img, h in ((ps, sc["small"], hs), (pl, sc["large"], hl)):
            hdr = fits.Header()
            for k, v in h.items():
                if not k.startswith("NAXIS"):
                    hdr[k] = v
            fits.HDUList([fits.PrimaryHDU(), fits.ImageHDU(data=img, header=hdr)]).writeto(p, overwrite=True)
        for key, path in (("hdr_small", ps), ("hdr_large", pl)):
            with fits.open(path) as f:
                h = f[-1].header
                out[f"{key}_{tag}"] = np.array(json.dumps(
                    {k: (h[k] if isinstance(h[k], (str, bool, int)) else float(h[k])) for k in h.keys()
                     if k not in ("XTENSION", "BITPIX", "PCOUNT", "GCOUNT", "NAXIS", "")}))
        return ps, pl

    out = {}
    big = np.load(sys.argv[1], allow_pickle=False)
    spice = np.load(sys.argv[2], allow_pickle=False)
    ps, pl = write("big", big)
    ps4, pl4 = write("cfg4", spice)
    out.update(fingerprint_big=big["fingerprint"], fingerprint_cfg4=spice["fingerprint"])

    def run(name, small, large, axes, lags, call, **ck):
        """axes: the full lag axes of the config (crval1, crval2, crota); lags: the sub-lattice, values of those axes."""
        t = time.time()
        A = Alignment(large_fov_known_pointing=large, small_fov_to_correct=small, lag_crval1=np.asarray(lags[0]),
                      lag_crval2=np.asarray(lags[1]), lag_cdelt1=None, lag_cdelt2=None,
                      lag_crota=None if lags[2] is None else np.asarray(lags[2]), parallelism=True, counts_cpu_max=8)
        c = getattr(A, "align_using_" + call)(return_type="corr", **ck)[:, :, 0, 0, :, 0]
        idx = [np.array([int(np.argmin(np.abs(ax - v))) for v in lg]) for ax, lg in zip(axes, lags) if lg is not None]
        for ax, lg, ii in zip(axes, lags, idx):
            assert np.allclose(ax[ii], lg, rtol=0, atol=1e-12)
        i3 = idx[2] if len(idx) > 2 else np.array([0])
        grid = np.stack(np.meshgrid(idx[0], idx[1], i3, indexing="ij"), axis=-1).reshape(-1, 3)
        out.setdefault(name + "_index", []).append(grid)
        out.setdefault(name + "_corr", []).append(c.reshape(-1))
        print(name, c.shape, "max", float(np.nanmax(c)), "nan", int(np.isnan(c).sum()), f"{time.time() - t:.1f} s", flush=True)

    # cfg2
    ax = np.arange(-30.0, 31.0, 1.0)
    run("cfg2", ps, pl, (ax, ax, None), (np.arange(-30.0, 31.0, 10.0), np.arange(-30.0, 31.0, 10.0), None), "helioprojective")
    run("cfg2", ps, pl, (ax, ax, None), ([16.0, 17.0, 18.0], [-10.0, -9.0, -8.0], None), "helioprojective")
    # cfg3
    ax = np.arange(-60.0, 61.0, 1.0)
    carr = dict(lonlims=[200.0, 300.0], latlims=[-20.0, 20.0], shape=[2048, 2048])
    run("cfg3", ps, pl, (ax, ax, None), (np.arange(-60.0, 61.0, 20.0), np.arange(-60.0, 61.0, 20.0), None), "carrington", **carr)
    run("cfg3", ps, pl, (ax, ax, None), ([16.0, 17.0, 18.0], [-10.0, -9.0, -8.0], None), "carrington", **carr)
    # cfg4 (lags in arcsec; the reference converts them to the header's degrees, alignment.py:819-837)
    ax = np.arange(-30.0, 31.0, 1.0)
    axr = np.round(np.arange(-10, 11) * 0.1, 10)
    run("cfg4", ps4, pl4, (ax, ax, axr), ([-30.0, -15.0, 0.0, 15.0, 30.0], [-30.0, -15.0, 0.0, 15.0, 30.0],
                                          [-1.0, -0.5, 0.0, 0.3, 1.0]), "helioprojective")
    run("cfg4", ps4, pl4, (ax, ax, axr), ([17.0], [-9.0], [0.2, 0.4]), "helioprojective")
    # cfg5: the d_cdelt = 0 plane
    ax = np.arange(-20.0, 21.0, 1.0)
    axr = np.round(np.arange(-5, 6) * 0.1, 10)
    run("cfg5", ps, pl, (ax, ax, axr), ([-20.0, -1.0, 17.0], [-9.0, 3.0, 20.0], [-0.5, 0.0, 0.3]), "carrington",
        lonlims=[200.0, 300.0], latlims=[-20.0, 20.0], shape=[4096, 4096])
    for k in list(out):
        if isinstance(out[k], list):
            out[k] = np.concatenate(out[k])
    for name in ("cfg2", "cfg3", "cfg4", "cfg5"):
        assert np.unique(out[name + "_index"], axis=0).shape[0] == out[name + "_index"].shape[0], name
    dst = os.path.join(HERE, "configs_reference.npz")
    np.savez(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes;", {n: int(out[n + "_corr"].size) for n in ("cfg2", "cfg3", "cfg4", "cfg5")})


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dump-cfg4-scene":
        dump_cfg4_scene(sys.argv[2])
        sys.exit(0)
    main()
