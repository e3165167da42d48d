"""GPU test of the SPICE caller against the RANDOM family the reference's own `AlignmentSpice` produced
(tests/golden/spice_fuzz_golden.{npz,json}; generator tests/golden/make_golden_spice_fuzz.py): every window end to end
-- L2 FITS file written by this package's writer, `_extract_spice_data_header`, the C ABI, the HIP sweep -- against the
reference's correlation map; 1e-7 (samples rounded to float32), identical NaN pattern and argmax."""
import warnings

import numpy as np
import pytest

from tests.test_reference_spice_fuzz_cpu import inputs, load, make_spice, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", scenes())
def test_alignment_spice_reproduces_the_reference_map(name, tmp_path):
    from euispice_coreg_amd.utils import fits_io
    g, m = load()
    cube, h4, large, hl = inputs(g, m, name)
    p_spice = str(tmp_path / m["scenes"][name]["file"])
    fits_io.write_images(p_spice, [(cube, h4)])
    p_large = str(tmp_path / "solo_L2_eui-fsi174-image_ref.fits")
    fits_io.write_images(p_large, [(None, {}), (large, hl)])
    A, c = make_spice(name, g, m, small=p_spice, large=p_large)
    A.level = None  # from the file name, as the reference (alignment_spice.py:94-98)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = A.align_using_helioprojective(return_type="corr", **c["call_kwargs"])
    want = g[f"{name}/corr"]
    assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-7, np.nanmax(np.abs(got - want))
    assert np.nanargmax(got) == np.nanargmax(want)
