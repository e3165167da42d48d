"""GPU tests of the callers either side of the sweep against what the REFERENCE's own classes produced
(tests/golden/callers_golden.{npz,json}; generator tests/golden/make_golden_callers.py):
`AlignmentSpice.align_using_helioprojective` end to end, `SPICEComposedMapBuilder.process` (the synthetic raster and its
header), `jitter_correction_imagers` (the corrected headers of a 4-image series) -- all through FITS files written by this
package's own writer, the C ABI and the HIP kernels."""
import os
import warnings

import numpy as np
import pytest

from tests.test_reference_callers_cpu import SPICE_CASES, load, make_spice, spice_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    from euispice_coreg_amd import synthetic
    from euispice_coreg_amd.utils import fits_io
    g, m = load()
    d = tmp_path_factory.mktemp("callers")
    cube, h4, large, hl = spice_inputs(g, m)
    p_spice = str(d / "solo_L2_spice-n-ras_20220317T094045_V01.fits")
    fits_io.write_images(p_spice, [(cube, h4)])
    p_large = str(d / "solo_L2_eui-fsi174-image_ref.fits")
    fits_io.write_images(p_large, [(None, {}), (large, hl)])
    imagers = []
    frames = synthetic.make_imager_sequence(large.astype(np.float64), hl)
    for k, (img, _) in enumerate(frames):
        p = str(d / f"solo_L2_eui-fsi174-image_{k:02d}.fits")
        fits_io.write_images(p, [(None, {}), (img, m["synras"]["imager_headers"][k])])
        imagers.append(p)
    series = []
    for k, h in enumerate(m["jitter"]["headers"]):
        p = str(d / f"solo_L2_eui-hrieuv174-image_{k:03d}.fits")
        fits_io.write_images(p, [(None, {}), (g[f"jitter/frame{k}"], h)])
        series.append(p)
    return dict(g=g, m=m, dir=d, spice=p_spice, large=p_large, imagers=imagers, series=series)


@pytest.mark.parametrize("case", SPICE_CASES)
def test_alignment_spice_reproduces_the_reference_map(files, case):
    g, m = files["g"], files["m"]
    A, c = make_spice(case, g, m, small=files["spice"], large=files["large"])
    A.level = None  # from the file name, as the reference (alignment_spice.py:94-98)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = A.align_using_helioprojective(return_type="corr", **c["call_kwargs"])
    want = g[f"spice/{case}/corr"]
    assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-7, np.nanmax(np.abs(got - want))
    assert np.nanargmax(got) == np.nanargmax(want)


def test_synthetic_raster_equals_the_references(files, tmp_path):
    """map_builder.py:87-214: every column of the raster sampled from the imager frame the reference chose, float32
    samples (interpol2d allocates the imager's dtype, Util.py:91-94) -- and the composed header, card for card."""
    from euispice_coreg_amd.synras.map_builder import SPICEComposedMapBuilder
    from euispice_coreg_amd.utils import fits_io
    g, m = files["g"], files["m"]
    for name, c in m["synras"]["cases"].items():
        if "header" not in c:
            continue
        C = SPICEComposedMapBuilder(path_to_spectro=files["spice"], list_imager_paths=files["imagers"],
                                    threshold_time=200.0, window_imager=-1, window_spectro=0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = C.process(folder_path_output=str(tmp_path), basename_output=f"{name}.fits", print_filename=False,
                            return_synras_name=True, **c["kwargs"])
        data, hdr = fits_io.read_image(out, 0)
        want = g[f"synras/{name}/raster"]
        assert list(data.shape) == c["shape"] and np.array_equal(np.isnan(data), np.isnan(want)), name
        # wcslib's coordinates against the exact homography: 1e-10 px, i.e. a float32 rounding flip now and then
        rel = np.nanmax(np.abs(data - want) / np.abs(want))
        assert rel <= 1.3e-7 and (data == want)[np.isfinite(want)].mean() > 0.999, (name, rel)
        for k, v in c["header"].items():
            if k in ("WCSAXES", "LATPOLE", "MJDREF", "MJD-OBS") or k.startswith("NAXIS"):
                continue
            assert k in hdr, (name, k)
            if isinstance(v, float):
                # (CRVAL of keep_original_imager_pixel_size comes back from a pixel -> world evaluation at the raster's
                # centre, map_builder.py:172-190: wcslib's chain against the package's TAN restatement, 1e-14 deg)
                assert hdr[k] == pytest.approx(v, rel=1e-14, abs=1e-12 if k.startswith("CRVAL") else 1e-300), (name, k, hdr[k], v)
            else:
                assert hdr[k] == v, (name, k, hdr[k], v)
    with pytest.raises(ValueError):  # map_builder.py:101-105
        SPICEComposedMapBuilder(files["spice"], files["imagers"][:2], threshold_time=100.0, window_spectro=0).process(
            folder_path_output=str(tmp_path), basename_output="x.fits", print_filename=False)


def test_alignment_on_the_synthetic_raster_equals_the_references(files, tmp_path):
    """README flow of the reference: build the synthetic raster, then align the SPICE window on it."""
    from euispice_coreg_amd.hdrshift import AlignmentSpice
    from euispice_coreg_amd.synras.map_builder import SPICEComposedMapBuilder
    g, m = files["g"], files["m"]
    C = SPICEComposedMapBuilder(files["spice"], files["imagers"], threshold_time=200.0, window_spectro=0)
    out = C.process(str(tmp_path), "synras.fits", print_filename=False, return_synras_name=True)
    c = m["synras"]["cases"]["align_on_raster"]
    A = AlignmentSpice(out, files["spice"], lag_crval1=np.asarray(c["lag_crval1"]), lag_crval2=np.asarray(c["lag_crval2"]),
                       large_fov_window=0, small_fov_window=0, parallelism=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = A.align_using_helioprojective(return_type="corr")
    want = g["synras/align_on_raster/corr"]
    # the raster itself carries float32 rounding flips against the reference's (above): 1e-6 on the coefficient
    assert np.nanmax(np.abs(got - want)) <= 1e-6 and np.nanargmax(got) == np.nanargmax(want)


def test_jitter_session_writes_the_references_headers(files, tmp_path):
    """jitter_correction.py:14-174: sublists, chain of corrected reference images, corrected FITS per image."""
    from euispice_coreg_amd.jitter_correction import jitter_correction_imagers
    from euispice_coreg_amd.utils import fits_io
    g, m = files["g"], files["m"]
    call = dict(m["jitter"]["call"])
    lag1, lag2 = np.asarray(call.pop("lag_crval1")), np.asarray(call.pop("lag_crval2"))
    out = str(tmp_path / "out")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        jitter_correction_imagers(files["series"], out, lag_crval1=lag1, lag_crval2=lag2, parallelism=True, **call)
    for k, (p, ref) in enumerate(zip(files["series"], m["jitter"]["outputs"])):
        o = os.path.join(out, os.path.basename(p))
        data, hdr = fits_io.read_image(o, -1)
        assert np.array_equal(data, g[f"jitter/frame{k}"], equal_nan=True) == ref["same_pixels"]
        if ref["byte_copy_of_input"]:
            assert open(o, "rb").read() == open(p, "rb").read()
        # scipy 1.7.1's curve_fit in the reference run, the library's restatement of scipy's TRF here: both stop within
        # 1e-3 px (2e-3 arcsec at this 2-arcsec lag step) of the same minimum; the error chains through the sublists
        assert abs(hdr["CRVAL1"] - ref["CRVAL1"]) < 5e-3 and abs(hdr["CRVAL2"] - ref["CRVAL2"]) < 5e-3, (k, hdr["CRVAL1"], ref)
        assert hdr["CROTA"] == ref["CROTA"] and hdr["PC1_1"] == pytest.approx(ref["PC1_1"], rel=1e-15)
        assert hdr["PC1_2"] == pytest.approx(ref["PC1_2"], rel=1e-15)
