"""GPU tests of the image hand-over: the reference-image crop (only the rectangle the target grid can touch crosses
PCIe) and the device-source entry points (replicas assembled by an all-gather) must give the very pixels and maps the
plain host uploads give."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

SHAPE = (72, 64)


@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("f32", [True, False])
def test_reference_crop_is_bit_identical(gpu_handle, order, f32):
    """coreg_prepare_reference_*: the prepared reference with the source image cropped to the bounding box of the sample
    coordinates (default) equals, bit for bit, the one prepared from the whole image -- Carrington grid (the limb cuts
    it: NaN outside), helioprojective sub-map, and a grid that reaches the image border (mirrored taps)."""
    from euispice_coreg_amd import _lib
    small, hs, large, hl, _ = H.scene(small_n=96, large_n=200)
    large = large.astype(np.float32) if f32 else large + 1e-9 * np.arange(large.size).reshape(large.shape)
    grids = [_lib.Grid(H.CARR_LON, H.CARR_LAT, SHAPE), _lib.Grid((150.0, 350.0), (-60.0, 60.0), (90, 80)),
             _lib.Grid((244.0, 246.5), (3.0, 5.0), (40, 50))]
    gpu_handle.set_small(small)
    for grid in grids:
        out = []
        for crop in (1, 0):
            gpu_handle.set_option("crop_reference", crop)
            try:
                gpu_handle.prepare_reference_carrington(large, hl, grid, 1.004, order)
                out.append(gpu_handle.get_reference_on_grid((grid.n_lat, grid.n_lon), np.float64))
            finally:
                gpu_handle.set_option("crop_reference", 1)
        assert np.array_equal(out[0], out[1], equal_nan=True)
    assert np.isfinite(out[0]).any()
    # helioprojective sub-map (small header's grid inside the large image), and a target that sticks out of the image
    hs_out = dict(hs)
    hs_out["CRVAL1"] = hl["CRVAL1"] + 0.5 * hl["NAXIS1"] * hl["CDELT1"]  # centred on the edge of the large image
    for hdr in (hs, hs_out):
        out = []
        for crop in (1, 0):
            gpu_handle.set_option("crop_reference", crop)
            try:
                gpu_handle.prepare_reference_helioprojective(large, hl, hdr, order)
                out.append(gpu_handle.get_reference_on_grid((hdr["NAXIS2"], hdr["NAXIS1"]), np.float32))
            finally:
                gpu_handle.set_option("crop_reference", 1)
        assert np.array_equal(out[0], out[1], equal_nan=True)
    assert np.isnan(out[0]).any() and np.isfinite(out[0]).any()  # the shifted target does leave the image


def test_full_size_reference_crop_and_upload_sizes(gpu_handle):
    """Headline inputs: the 2048 x 2048 Carrington grid touches a small part of the 3072 x 3072 reference; cropped and
    whole uploads prepare the same reference, and the sweep over it finds the injected shift."""
    from euispice_coreg_amd import _lib, synthetic
    small, hs, large, hl, truth = synthetic.make_scene()
    large32 = large.astype(np.float32)
    grid = _lib.Grid((200.0, 300.0), (-20.0, 20.0), (2048, 2048))
    ref = []
    for crop in (1, 0):
        gpu_handle.set_option("crop_reference", crop)
        try:
            gpu_handle.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
            ref.append(gpu_handle.get_reference_on_grid((2048, 2048), np.float64))
        finally:
            gpu_handle.set_option("crop_reference", 1)
    assert np.array_equal(ref[0], ref[1], equal_nan=True)
    gpu_handle.set_small(small.astype(np.float32))
    gpu_handle.prepare_reference_carrington(large32, hl, grid, 1.004, 2)
    lags = _lib.LagSet(np.arange(13.0, 22.0), np.arange(-13.0, -4.0), None, None, None)
    corr = gpu_handle.sweep_carrington(hs, grid, 1.004, lags).reshape(9, 9)
    assert np.unravel_index(np.argmax(corr), corr.shape) == (4, 4)


def test_device_source_uploads_equal_host_uploads(gpu_handle):
    """coreg_set_small_from_device / coreg_prepare_reference_*_from_device (the multi-GPU hand-over: replicas assembled
    on the device by an all-gather) against the host uploads: same resident pixels, same maps; float32 and float64."""
    import torch
    from euispice_coreg_amd import _lib
    small, hs, large, hl, _ = H.scene()
    grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, SHAPE)
    lags = _lib.LagSet(17.0 + 2.0 * (np.arange(5) - 2), -9.0 + 2.0 * (np.arange(4) - 2), None, None, [0.0, 0.3])
    noisy = small + 1e-7 * np.arange(small.size).reshape(small.shape)  # not float32-exact: stays float64 on the device
    for s_img, l_img in ((small.astype(np.float32), large.astype(np.float32)), (noisy, large)):
        gpu_handle.set_small(s_img)
        gpu_handle.prepare_reference_carrington(l_img, hl, grid, 1.004, 2)
        want_c = gpu_handle.sweep_carrington(hs, grid, 1.004, lags)
        gpu_handle.prepare_reference_helioprojective(l_img, hl, hs, 2)
        want_h = gpu_handle.sweep_helioprojective(hs, hs, lags)
        ts, tl = torch.from_numpy(np.ascontiguousarray(s_img)).cuda(), torch.from_numpy(np.ascontiguousarray(l_img)).cuda()
        torch.cuda.synchronize()
        gpu_handle.set_small_from_device(ts.data_ptr(), ts.shape, s_img.dtype)
        gpu_handle.prepare_reference_carrington_from_device(tl.data_ptr(), tl.shape, l_img.dtype, hl, grid, 1.004, 2)
        got_c = gpu_handle.sweep_carrington(hs, grid, 1.004, lags)
        gpu_handle.prepare_reference_helioprojective_from_device(tl.data_ptr(), tl.shape, l_img.dtype, hl, hs, 2)
        got_h = gpu_handle.sweep_helioprojective(hs, hs, lags)
        gpu_handle.synchronize()
        assert np.array_equal(got_c, want_c, equal_nan=True) and np.array_equal(got_h, want_h, equal_nan=True)
        assert gpu_handle.last_stats()["small_is_f32"] == int(s_img.dtype == np.float32)
