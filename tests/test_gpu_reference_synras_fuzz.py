"""GPU test of `SPICEComposedMapBuilder.process` against the RANDOM family the reference's own builder produced
(tests/golden/synras_fuzz_golden.{npz,json}; generator tests/golden/make_golden_synras_fuzz.py): four random SPICE
windows x imager sequences of 5 to 17 frames with random start and cadence, with and without
`keep_original_imager_pixel_size`.  map_builder.py:87-214: the raster (float32 samples of the frame the reference chose
for every column) and the composed header, card for card."""
import warnings

import numpy as np
import pytest

from tests.test_reference_spice_fuzz_cpu import inputs
from tests.test_reference_spice_fuzz_cpu import load as load_spice
from tests.test_reference_synras_fuzz_cpu import load, windows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", windows())
def test_synthetic_raster_equals_the_references(name, tmp_path):
    from euispice_coreg_amd import synthetic
    from euispice_coreg_amd.synras.map_builder import SPICEComposedMapBuilder
    from euispice_coreg_amd.utils import fits_io
    g, m, sp = load()
    gs, ms = load_spice()
    c = m["cases"][name]
    cube, h4, large, hl = inputs(gs, ms, name)
    p_spice = str(tmp_path / sp[name]["file"])
    fits_io.write_images(p_spice, [(cube, h4)])
    frames = synthetic.make_imager_sequence(large.astype(np.float64), hl, start=c["start"], cadence_s=c["cadence_s"],
                                            n_frames=c["n_frames"])
    paths = []
    for j, (img, _) in enumerate(frames):
        p = str(tmp_path / f"solo_L2_eui-fsi174-image_{j:02d}.fits")
        fits_io.write_images(p, [(None, {}), (img, c["imager_headers"][j])])
        paths.append(p)
    C = SPICEComposedMapBuilder(path_to_spectro=p_spice, list_imager_paths=paths, threshold_time=c["threshold_time"],
                                window_imager=-1, window_spectro=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = C.process(folder_path_output=str(tmp_path), basename_output="synras.fits", print_filename=False,
                        return_synras_name=True, **c["kwargs"])
    data, hdr = fits_io.read_image(out, 0)
    want = g[f"{name}/raster"]
    assert list(data.shape) == c["shape"] and np.array_equal(np.isnan(data), np.isnan(want))
    # wcslib's coordinates against the exact homography: 1e-10 px, i.e. a float32 rounding flip now and then
    rel = np.nanmax(np.abs(data - want) / np.abs(want))
    assert rel <= 1.3e-7 and (data == want)[np.isfinite(want)].mean() > 0.999, rel
    for k, v in c["header"].items():
        if k in ("WCSAXES", "LATPOLE", "MJDREF", "MJD-OBS") or k.startswith("NAXIS"):
            continue
        assert k in hdr, k
        if isinstance(v, float):
            assert hdr[k] == pytest.approx(v, rel=1e-14, abs=1e-12 if k.startswith("CRVAL") else 1e-300), (k, hdr[k], v)
        else:
            assert hdr[k] == v, (k, hdr[k], v)
