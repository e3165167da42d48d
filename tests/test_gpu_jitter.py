"""GPU tests of the jitter-correction session (euispice_coreg_amd.jitter_correction, reference:
jitter_correction/jitter_correction.py:14-174) on a synthetic jittering series written as FITS files."""
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

LON, LAT, SHAPE = (236.0, 256.0), (-4.0, 16.0), (200, 200)


@pytest.fixture(scope="module")
def series(tmp_path_factory):
    from euispice_coreg_amd import synthetic
    from euispice_coreg_amd.utils import fits_io
    d = tmp_path_factory.mktemp("series")
    frames, jit = synthetic.make_series(n_frames=5, n=256, seed=7, jitter_sigma=4.0)
    paths = []
    for k, (img, hdr) in enumerate(frames):
        p = str(d / f"solo_L2_eui-hrieuv174-image_{k:03d}.fits")
        fits_io.write_images(p, [(None, {}), (img, hdr)])
        paths.append(p)
    return paths, frames, jit, d


def test_jitter_series_matches_oracle_chain(series, tmp_path):
    """Every output header = input header + the shift the ORACLE's sweep finds against the same (corrected) reference,
    following the sublist chain of the reference implementation; and the recovered pointing is the injected jitter."""
    from euispice_coreg_amd.jitter_correction import jitter_correction_imagers
    from euispice_coreg_amd.hdrshift import AlignmentResults
    from euispice_coreg_amd.utils import fits_io
    paths, frames, jit, _ = series
    out = str(tmp_path / "out")
    lag = np.arange(-12.0, 12.5, 1.0)
    figs = str(tmp_path / "figs")
    done = jitter_correction_imagers(paths, out, lonlims=LON, latlims=LAT, shape=SHAPE, lag_crval1=lag, lag_crval2=lag,
                                     sublist_length=2, overlap=1, small_fov_value_max=2800.0, path_figures=figs)
    assert len([f for f in os.listdir(figs) if f.startswith("correlation_") and f.endswith(".pdf")]) == 4
    assert [(a, r) for a, r, _ in done] == [(1, 0), (2, 0), (3, 2), (4, 2)]
    outs = [os.path.join(out, os.path.basename(p)) for p in paths]
    assert all(os.path.isfile(p) for p in outs)
    # frame 0 is copied untouched
    assert open(outs[0], "rb").read() == open(paths[0], "rb").read()

    corrected = {0: dict(frames[0][1])}
    for idx, ref, res in done:
        img = frames[idx][0].astype(np.float64)
        from oracle import coreg_oracle as O
        O.set_threshold_minmax_to_nan(img, None, 2800.0)
        ref_img = frames[ref][0].astype(np.float64)
        want = H.oracle_carrington(img, frames[idx][1], ref_img, corrected[ref], (lag, lag, [0], [0], [0]), SHAPE, LON,
                                   LAT)
        H.assert_corr_close(res.corr, want, 1e-10, f"jitter frame {idx} vs {ref}")
        r = AlignmentResults(want, lag, lag, [0], [0], [0], "arcsec")
        hdr_out = fits_io.read_header(outs[idx], -1)
        # the Gaussian sub-lag fit amplifies the 1e-10 differences of the maps: SURVEY 8d gate, 1e-3 arcsec
        assert abs(hdr_out["CRVAL1"] - (frames[idx][1]["CRVAL1"] + r.shift_arcsec[0])) < 1e-3
        assert abs(hdr_out["CRVAL2"] - (frames[idx][1]["CRVAL2"] + r.shift_arcsec[1])) < 1e-3
        corrected[idx] = dict(frames[idx][1], CRVAL1=hdr_out["CRVAL1"], CRVAL2=hdr_out["CRVAL2"])
        # pixels are carried over unchanged
        d_out, _ = fits_io.read_image(outs[idx], -1)
        assert np.array_equal(d_out, frames[idx][0], equal_nan=True)
        # the injected jitter is recovered (pixel scale 3.9 arcsec, lag step 1 arcsec)
        assert abs(hdr_out["CRVAL1"] - (frames[idx][1]["CRVAL1"] + jit[idx, 0])) < 1.0
        assert abs(hdr_out["CRVAL2"] - (frames[idx][1]["CRVAL2"] + jit[idx, 1])) < 1.0


def test_reference_stays_resident_within_a_sublist(series, tmp_path, monkeypatch):
    """One reference preparation per sublist, not per image."""
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.jitter_correction import jitter_correction_imagers
    paths = series[0]
    calls = []
    orig = _lib.CoregHandle.prepare_reference_carrington

    def spy(self, *a, **k):
        calls.append(1)
        return orig(self, *a, **k)
    monkeypatch.setattr(_lib.CoregHandle, "prepare_reference_carrington", spy)
    lag = np.arange(-2.0, 2.5, 1.0)
    jitter_correction_imagers(paths, str(tmp_path / "o"), lonlims=LON, latlims=LAT, shape=SHAPE, lag_crval1=lag,
                              lag_crval2=lag, sublist_length=4, overlap=1, pipeline_depth=1)
    assert len(calls) == 1  # sublists [0..4] and [4]: four sweeps, one preparation
    del calls[:]
    a = jitter_correction_imagers(paths, str(tmp_path / "o2"), lonlims=LON, latlims=LAT, shape=SHAPE, lag_crval1=lag,
                                  lag_crval2=lag, sublist_length=4, overlap=1, pipeline_depth=2)
    assert len(calls) <= 2  # one per library context
    b = jitter_correction_imagers(paths, str(tmp_path / "o3"), lonlims=LON, latlims=LAT, shape=SHAPE, lag_crval1=lag,
                                  lag_crval2=lag, sublist_length=4, overlap=1, pipeline_depth=1)
    # the schedule does not change the numbers
    assert [i for i, _, _ in a] == [i for i, _, _ in b]
    assert all(np.array_equal(x[2].corr, y[2].corr, equal_nan=True) for x, y in zip(a, b))


def test_device_thresholds_equal_host_thresholds(series):
    """coreg_set_small_f32 + coreg_threshold_small == host-side masking + coreg_set_small, bit for bit."""
    from euispice_coreg_amd import _lib
    from oracle import coreg_oracle as O
    frames = series[1]
    img32, hdr = frames[1]
    ref = frames[0][0].astype(np.float64)
    lag = np.arange(-3.0, 3.5, 1.5)
    lags = (lag, lag, None, None, None)
    h = _lib.shared_handle(-1)
    host = img32.astype(np.float64)
    O.set_threshold_minmax_to_nan(host, 150.0, 2500.0)
    a = H.gpu_carrington(h, host, hdr, ref, frames[0][1], lags, SHAPE, LON, LAT)
    grid = _lib.Grid(LON, LAT, SHAPE)
    h.set_small(img32)
    n = h.threshold_small(150.0, 2500.0)
    assert n == int(np.isfinite(host).sum())
    b = h.sweep_carrington(hdr, grid, 1.004, _lib.LagSet(*lags)).reshape(a.shape)
    assert np.array_equal(a, b, equal_nan=True)
    # float64 storage path (values not representable in float32)
    img64 = img32.astype(np.float64) * (1.0 + 1e-12)
    host = img64.copy()
    O.set_threshold_minmax_to_nan(host, 150.0, 2500.0)
    a = H.gpu_carrington(h, host, hdr, ref, frames[0][1], lags, SHAPE, LON, LAT)
    h.set_small(img64)
    assert h.threshold_small(150.0, 2500.0) == int(np.isfinite(host).sum())
    b = h.sweep_carrington(hdr, grid, 1.004, _lib.LagSet(*lags)).reshape(a.shape)
    assert np.array_equal(a, b, equal_nan=True)
    with pytest.raises(ValueError):
        from euispice_coreg_amd.hdrshift import Alignment
        Alignment((ref, frames[0][1]), (img32, hdr), lag, lag, None, None, None,
                  small_fov_value_min=1e9).align_using_carrington(lonlims=LON, latlims=LAT, shape=SHAPE)


def test_two_library_contexts_sweep_concurrently(series):
    """Two host threads, each with its own handle (own stream and buffers) on the same GPU, run different sweeps at the
    same time -- the jitter session's driver threads do -- and get the results of the sequential runs, bit for bit."""
    import threading
    from euispice_coreg_amd import _lib
    frames = series[1]
    ref = frames[0][0].astype(np.float64)
    grid = _lib.Grid(LON, LAT, SHAPE)
    lagsets = [_lib.LagSet(np.arange(-6.0, 6.5, 1.0) + 0.25 * k, np.arange(-5.0, 5.5, 1.0), None, None, [0.0, 0.2 * k])
               for k in range(1, 5)]

    def run(h, k, out):
        h.set_small(frames[k][0])
        h.threshold_small(None, 2800.0)
        out[k] = h.sweep_carrington(frames[k][1], grid, 1.004, lagsets[k - 1])

    hs = [_lib.shared_handle(-1, slot=0), _lib.shared_handle(-1, slot=1)]
    for h in hs:
        h.prepare_reference_carrington(ref, frames[0][1], grid, 1.004, 2)
    seq, par = {}, {}
    for k in range(1, 5):
        run(hs[0], k, seq)
    for rounds in range(3):  # several rounds: the interleaving differs every time
        th = [threading.Thread(target=lambda i=i: [run(hs[i], k, par) for k in range(1 + i, 5, 2)]) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k in range(1, 5):
            assert np.array_equal(seq[k], par[k], equal_nan=True), (rounds, k)


def test_session_drives_every_device_from_one_process(series, tmp_path, monkeypatch):
    """The plain-script session on several GPUs (here: two logical devices on the one GPU, COREG_VIRTUAL_DEVICES=2): the
    images of a sublist are dealt to the devices, each with its own contexts and its own prepared reference; corrected
    headers, correlation maps and files equal the one-device session's; `device=` / COREG_SINGLE_DEVICE=1 opt out."""
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.jitter_correction import jitter_correction_imagers
    from euispice_coreg_amd.jitter_correction.jitter_correction import session_devices
    from euispice_coreg_amd.utils import fits_io
    paths = series[0]
    lag = np.arange(-12.0, 12.5, 1.0)
    kw = dict(lonlims=LON, latlims=LAT, shape=SHAPE, lag_crval1=lag, lag_crval2=lag, sublist_length=4, overlap=1,
              small_fov_value_max=2800.0)
    one = jitter_correction_imagers(paths, str(tmp_path / "one"), device=0, **kw)
    assert session_devices(None, 1) == [(-1, 0)] and session_devices(3, 1) == [(3, 0)]
    monkeypatch.setenv("COREG_VIRTUAL_DEVICES", "2")
    assert session_devices(None, 1) == [(0, 0), (0, 16)]
    monkeypatch.setenv("COREG_SINGLE_DEVICE", "1")
    assert session_devices(None, 1) == [(-1, 0)]
    monkeypatch.delenv("COREG_SINGLE_DEVICE")
    used = []
    orig = _lib.shared_handle

    def spy(device=-1, slot=0):
        used.append((int(device), int(slot)))
        return orig(device, slot)
    monkeypatch.setattr(_lib, "shared_handle", spy)
    two = jitter_correction_imagers(paths, str(tmp_path / "two"), **kw)
    assert {s // 16 for _, s in used} == {0, 1}            # both logical devices swept
    assert [(i, r) for i, r, _ in one] == [(i, r) for i, r, _ in two]
    for (_, _, a), (_, _, b) in zip(one, two):
        assert np.array_equal(a.corr, b.corr, equal_nan=True) and a.shift_arcsec == b.shift_arcsec
    for p in paths:
        f1, f2 = (os.path.join(str(tmp_path / d), os.path.basename(p)) for d in ("one", "two"))
        assert open(f1, "rb").read() == open(f2, "rb").read()
        assert fits_io.read_header(f1, -1)["CRVAL1"] == fits_io.read_header(f2, -1)["CRVAL1"]


def test_session_on_a_tile_compressed_series(series, tmp_path):
    """The series as EUI distributes it -- every image a Rice-compressed HDU (written here by
    fits_io.write_compressed_image, lossless 16-bit integers as in level-1 files): the session uploads the compressed
    bytes, the GPU decodes them, and the corrected files keep the compressed stream.  Same shifts as the session on the
    plain files holding the same pixels."""
    from euispice_coreg_amd.jitter_correction import jitter_correction_imagers
    from euispice_coreg_amd.utils import fits_io
    paths, frames, jit, _ = series
    plain_dir, comp_dir = tmp_path / "plain", tmp_path / "comp"
    plain_dir.mkdir()
    comp_dir.mkdir()
    plain, comp = [], []
    for k, (img, hdr) in enumerate(frames):
        px = np.clip(np.nan_to_num(img, nan=0.0) * 16.0, 0, 65535).astype(np.uint16)
        name = f"solo_L1_eui-hrieuv174-image_{k:03d}.fits"
        fits_io.write_images(str(plain_dir / name), [(None, {}), (px.astype(np.float32), hdr)])
        fits_io.write_compressed_image(str(comp_dir / name), px, hdr)
        plain.append(str(plain_dir / name))
        comp.append(str(comp_dir / name))
        assert fits_io.open_compressed(comp[-1], -1).on_gpu
    lag = np.arange(-12.0, 12.5, 1.0)
    kw = dict(lonlims=LON, latlims=LAT, shape=SHAPE, lag_crval1=lag, lag_crval2=lag, sublist_length=2, overlap=1)
    done_p = jitter_correction_imagers(plain, str(tmp_path / "out_p"), **kw)
    done_c = jitter_correction_imagers(comp, str(tmp_path / "out_c"), **kw)
    assert [(a, r) for a, r, _ in done_p] == [(a, r) for a, r, _ in done_c]
    for (_, _, rp), (_, _, rc) in zip(done_p, done_c):
        assert np.array_equal(rp.corr, rc.corr, equal_nan=True)
    for k, p in enumerate(comp):
        o = os.path.join(str(tmp_path / "out_c"), os.path.basename(p))
        ci_in, ci_out = fits_io.open_compressed(p, -1), fits_io.open_compressed(o, -1)
        assert ci_out is not None and ci_out.on_gpu, "the corrected file is tile-compressed like its input"
        assert bytes(ci_out._heap) == bytes(ci_in._heap)
        hp = fits_io.read_header(os.path.join(str(tmp_path / "out_p"), os.path.basename(p)), -1)
        assert ci_out.header["CRVAL1"] == hp["CRVAL1"] and ci_out.header["CRVAL2"] == hp["CRVAL2"]
        if k:
            assert abs(ci_out.header["CRVAL1"] - (frames[k][1]["CRVAL1"] + jit[k, 0])) < 1.0
