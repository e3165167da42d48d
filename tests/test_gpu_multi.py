"""GPU tests of the in-library multi-GPU driver (include/coreg_hip.h: coreg_multi; euispice_coreg_amd/csrc/multi.hpp): every
GPU of the node from ONE process, the way the reference's `parallelism=True` uses the whole machine from a plain script
(hdrshift/alignment.py:692-744).  The GPU box has one GPU: COREG_VIRTUAL_DEVICES maps several logical devices onto it
(own host thread, library context and stream each), the blocks then reach the host by one copy per device instead of
the RCCL all-gather (RCCL refuses two ranks on one device); the RCCL calls themselves run with a one-rank group."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

SHAPE = (72, 64)


def _single_carr(h, small, hs, large, hl, lags, **kw):
    return H.gpu_carrington(h, small, hs, large, hl, lags, SHAPE, **kw).ravel()


def _multi_carr(m, small, hs, large, hl, lags, order=2, method=0):
    from euispice_coreg_amd import _lib
    grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, SHAPE)
    m.set_small(small)
    m.prepare_reference_carrington(large, hl, grid, 1.004, order)
    return m.sweep_carrington(hs, grid, 1.004, _lib.LagSet(*lags), order=order, method=method)


@pytest.mark.parametrize("n_virtual", [2, 3, 4])
def test_multi_every_partition_equals_single_device(gpu_handle, monkeypatch, n_virtual):
    """Blocks of the lag plane, runs of (cdelt1, cdelt2, crota) combinations (alone and mixed with blocks), raveled slices
    and grid shares: the planner's choice for each lag set, and every lag partition forced on the 3-D lag set."""
    from euispice_coreg_amd import _lib, parallel
    monkeypatch.setenv("COREG_VIRTUAL_DEVICES", str(n_virtual))
    small, hs, large, hl, _ = H.scene()
    assert _lib.device_count() == n_virtual
    l24, l23 = 17.0 + 1.0 * (np.arange(24) - 12), -9.0 + 1.0 * (np.arange(23) - 11)
    cases = [
        ("plane", (l24, l23, None, None, [0.3]), -1),
        ("3-D", (l24, l23, None, None, [0.0, 0.3, -0.2]), -1),
        ("3-D", (l24, l23, None, None, [0.0, 0.3, -0.2]), 1),
        ("3-D", (l24, l23, None, None, [0.0, 0.3, -0.2]), 2),
        ("3-D", (l24, l23, None, None, [0.0, 0.3, -0.2]), 4),
        ("5-D", (l24[:12], l23[:9], [0.0, 0.002], [-0.001, 0.0, 0.001], [0.0, 0.3]), -1),
        ("inner only", ([17.0], [-9.0], [0.0, 0.002], [-0.001, 0.0, 0.001], np.linspace(-1.0, 1.0, 90)), -1),
        ("inner only", ([17.0], [-9.0], [0.0, 0.002], [-0.001, 0.0, 0.001], np.linspace(-1.0, 1.0, 90)), 2),
        ("few", (17.0 + 2.0 * (np.arange(5) - 2), -9.0 + 2.0 * (np.arange(4) - 2), None, None, None), -1),
    ]
    seen = set()
    with _lib.MultiHandle() as m:
        assert m.size == n_virtual and m.collective == "host-copy" and m.rccl_status == "not used"
        for name, lags, force in cases:
            ls = _lib.LagSet(*lags)
            m.set_option("force_mode", force)
            got = _multi_carr(m, small, hs, large, hl, lags)
            mode = {-1: parallel.lag_sharding(ls.shape, n_virtual), 1: "blocks", 2: "slices", 4: "combos"}[force]
            assert m.last_mode == mode, (name, force, mode, m.last_mode)
            seen.add(mode)
            want = _single_carr(gpu_handle, small, hs, large, hl, lags)
            assert np.array_equal(np.isnan(got), np.isnan(want)), (name, mode)
            # (different lag sets cull the grid differently: agreement to rounding, as for slices of one handle)
            assert np.nanmax(np.abs(got - want)) <= 1e-12, (name, mode, n_virtual)
            assert np.nanargmax(got) == np.nanargmax(want)
            st = m.last_stats()
            assert st["n_devices"] == n_virtual and st["lag_sharding"] == mode
            assert len(st["per_device_sweep_kernel_ms"]) == n_virtual
        m.set_option("force_mode", -1)
        assert seen == {"blocks", "combos", "slices", "points"}
        # method 'residus' through the grid-share mode (sums from different devices add up about the same pivots)
        few = cases[-1][1]
        got = _multi_carr(m, small, hs, large, hl, few, method=1)
        grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, SHAPE)
        gpu_handle.set_small(small)
        gpu_handle.prepare_reference_carrington(large, hl, grid, 1.004, 2)
        want = gpu_handle.sweep_carrington(hs, grid, 1.004, _lib.LagSet(*few), method=1)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        if np.isfinite(want).any():
            assert np.nanmax(np.abs(got - want)) <= 1e-10
        # a sweep that fails inside the grid-share branch must not leave the contexts sharded: the next
        # lag-sharded sweep on the same handle gives the whole map
        with pytest.raises(_lib.CoregError):  # (refused by every device's sweep call, inside the grid-share branch)
            m.sweep_carrington(hs, grid, 1.004, _lib.LagSet(*few), order=9)
        assert m.last_mode == "points"
        got = _multi_carr(m, small, hs, large, hl, cases[0][1])
        want = _single_carr(gpu_handle, small, hs, large, hl, cases[0][1])
        assert m.last_mode == "blocks" and np.nanmax(np.abs(got - want)) <= 1e-12
        assert np.array_equal(np.isnan(got), np.isnan(want))


def test_combo_range_option_is_one_shot_and_selects_the_combinations(gpu_handle):
    """coreg_set_option "combo_begin" / "combo_end" on a single context: the next sweep covers that run of (cdelt1,
    cdelt2, crota) combinations only, output [n1][n2][run]; the sweep after it covers everything again; bad ranges are
    refused and leave nothing behind.  Both frames."""
    from euispice_coreg_amd import _lib
    small, hs, large, hl, _ = H.scene()
    lags = (17.0 + 1.0 * (np.arange(7) - 3), -9.0 + 1.0 * (np.arange(6) - 3), [0.0, 0.002], [-0.001, 0.0, 0.001],
            [0.0, 0.3, -0.2])
    ls = _lib.LagSet(*lags)
    inner = 2 * 3 * 3
    grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, SHAPE)
    gpu_handle.set_small(small)
    gpu_handle.prepare_reference_carrington(large, hl, grid, 1.004, 2)
    full = gpu_handle.sweep_carrington(hs, grid, 1.004, ls).reshape(7, 6, inner)
    for c_lo, c_hi in ((0, 5), (5, 6), (7, 18), (0, 18)):
        gpu_handle.set_option("combo_begin", c_lo)
        gpu_handle.set_option("combo_end", c_hi)
        part = gpu_handle.sweep_carrington(hs, grid, 1.004, ls, lag_end=7 * 6 * (c_hi - c_lo)).reshape(7, 6, c_hi - c_lo)
        assert np.array_equal(np.isnan(part), np.isnan(full[:, :, c_lo:c_hi]))
        assert np.nanmax(np.abs(part - full[:, :, c_lo:c_hi])) <= 1e-12
    again = gpu_handle.sweep_carrington(hs, grid, 1.004, ls).reshape(7, 6, inner)  # the range was consumed
    assert np.array_equal(again, full, equal_nan=True)
    gpu_handle.set_option("combo_begin", 4)
    gpu_handle.set_option("combo_end", 19)
    with pytest.raises(_lib.CoregError):
        gpu_handle.sweep_carrington(hs, grid, 1.004, ls, lag_end=10)
    again = gpu_handle.sweep_carrington(hs, grid, 1.004, ls).reshape(7, 6, inner)
    assert np.array_equal(again, full, equal_nan=True)
    #  a sweep that fails BEFORE the lag set is looked at -- a spline order the library refuses -- consumes the
    # range all the same: the next, ordinary sweep on this context covers everything
    gpu_handle.set_option("combo_begin", 2)
    gpu_handle.set_option("combo_end", 4)
    with pytest.raises(_lib.CoregError):
        gpu_handle.sweep_carrington(hs, grid, 1.004, ls, order=9)
    again = gpu_handle.sweep_carrington(hs, grid, 1.004, ls).reshape(7, 6, inner)
    assert np.array_equal(again, full, equal_nan=True)
    # helioprojective sub-map semantics, a slice of the restricted index
    gpu_handle.prepare_reference_helioprojective(large, hl, hs, 2)
    fullh = gpu_handle.sweep_helioprojective(hs, hs, ls).reshape(7, 6, inner)
    gpu_handle.set_option("combo_begin", 3)
    gpu_handle.set_option("combo_end", 11)
    part = gpu_handle.sweep_helioprojective(hs, hs, ls, lag_begin=8, lag_end=7 * 6 * 8 - 5)
    want = fullh[:, :, 3:11].ravel()[8:7 * 6 * 8 - 5]
    assert np.array_equal(np.isnan(part), np.isnan(want)) and np.nanmax(np.abs(part - want)) <= 1e-12


def test_multi_helioprojective_with_the_zero_lag(gpu_handle, monkeypatch):
    """Sub-map semantics over three logical devices, lag axes through 0: the noise-decided zero lag lands in ONE
    device's block and carries its border correction there."""
    from euispice_coreg_amd import _lib
    monkeypatch.setenv("COREG_VIRTUAL_DEVICES", "3")
    small, hs, large, hl, _ = H.scene()
    lags = (np.arange(-8.0, 20.0, 1.0), np.arange(-14.0, 3.0, 1.0), None, None, None)  # 28 x 17 = 476 lag-points
    want = H.gpu_helio(gpu_handle, small, hs, large, hl, lags).ravel()
    with _lib.MultiHandle() as m:
        m.set_small(small)
        m.prepare_reference_helioprojective(large, hl, hs, 2)
        got = m.sweep_helioprojective(hs, hs, _lib.LagSet(*lags))
        assert m.last_mode == "blocks"
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-12
    oracle = H.oracle_helio(small, hs, large, hl, ([0.0, 17.0], [-9.0, 0.0], None, None, None)).ravel()
    full = got.reshape(28, 17)
    assert abs(full[8, 14] - oracle[1]) <= 1e-7 and abs(full[25, 5] - oracle[2]) <= 1e-7  # (0, 0) and (17, -9)


def test_multi_rccl_calls_run_with_a_one_rank_group(gpu_handle, monkeypatch, tmp_path):
    """The RCCL side of the driver on the one GPU there is: dlopen, ncclCommInitAll, ncclGroupStart / ncclAllGather /
    ncclGroupEnd on the handle's stream, device-0 hand-over -- a one-rank group (the all-gather is then a copy)."""
    from euispice_coreg_amd import _lib
    monkeypatch.delenv("COREG_VIRTUAL_DEVICES", raising=False)
    monkeypatch.setenv("COREG_MULTI_FORCE_RCCL", "1")
    small, hs, large, hl, _ = H.scene()
    lags = (17.0 + 1.0 * (np.arange(12) - 6), -9.0 + 1.0 * (np.arange(12) - 6), None, None, None)
    with _lib.MultiHandle(device_ids=[0]) as m:
        if m.collective != "rccl":
            pytest.skip("no RCCL runtime could be loaded in this process")
        assert m.rccl_status == "ok"   # the communicator gathered the self-test pattern at creation
        got = _multi_carr(m, small, hs, large, hl, lags)
        assert m.last_mode == "slices" and m.collective == "rccl"
        want = _single_carr(gpu_handle, small, hs, large, hl, lags)
        assert np.array_equal(got, want, equal_nan=True)
        # the image to align by row shares + ONE all-gather (in place), against the whole-image hand-over: float32
        # pixels, float64 pixels that are not float32-exact, and the raw bytes of a FITS data unit
        from euispice_coreg_amd.utils import fits_io
        p = str(tmp_path / "s.fits")
        fits_io.write_images(p, [(None, {}), (small.astype(np.float32), hs)])
        noisy = small + 1e-7 * np.arange(small.size).reshape(small.shape)
        for img in (small.astype(np.float32), noisy, fits_io.open_raw(p, -1)):
            maps = []
            for shares in (1, 0):
                m.set_option("image_shares", shares)
                maps.append(_multi_carr(m, img, hs, large, hl, lags))
            assert np.array_equal(maps[0], maps[1], equal_nan=True)
            assert np.array_equal(maps[0], _single_carr(gpu_handle, np.asarray(img), hs, large, hl, lags), equal_nan=True)
        m.set_option("image_shares", 1)


def test_a_collective_that_never_completes_is_aborted_and_the_map_still_comes_out_right(gpu_handle, monkeypatch):
    """every RCCL wait of the one-process driver is bounded.  COREG_RCCL_TEST_STALL=1 puts, in place
    of the group's all-gather, a kernel on the stream that does not end before it is told to (the stream then looks
    exactly like one holding a collective that never completes); COREG_RCCL_WAIT_SECONDS bounds the wait.  On expiry the
    communicator is aborted, RCCL dropped for the handle, and the intact per-device block reaches the host by a copy:
    the right map, this sweep and the next.  Likewise a communicator bootstrap that does not return in time
    (COREG_RCCL_TEST_INIT_STALL_SECONDS) costs the handle RCCL, not the process."""
    import time
    from euispice_coreg_amd import _lib
    monkeypatch.delenv("COREG_VIRTUAL_DEVICES", raising=False)
    monkeypatch.setenv("COREG_MULTI_FORCE_RCCL", "1")
    small, hs, large, hl, _ = H.scene()
    lags = (17.0 + 1.0 * (np.arange(12) - 6), -9.0 + 1.0 * (np.arange(12) - 6), None, None, None)
    want = _single_carr(gpu_handle, small, hs, large, hl, lags)
    with _lib.MultiHandle(device_ids=[0]) as m:
        if m.collective != "rccl":
            pytest.skip("no RCCL runtime could be loaded in this process")
        monkeypatch.setenv("COREG_RCCL_TEST_STALL", "1")
        monkeypatch.setenv("COREG_RCCL_WAIT_SECONDS", "0.4")
        t0 = time.perf_counter()
        got = _multi_carr(m, small, hs, large, hl, lags)
        dt = time.perf_counter() - t0
        assert np.array_equal(got, want, equal_nan=True)
        assert "did not complete" in m.collective and "did not complete in time" in m.rccl_status
        # where the time of the recovery goes is reported by the library -- the poll of the streams
        # (the limit), ncclCommAbort (RCCL's own teardown), the drain of the streams after the release -- and the stall
        # kernel's self-limit is 12 s now: a recovery that only came back because the kernel gave up cannot pass
        import re
        ph = re.search(r"waited ([0-9.]+) s on the streams \(limit ([0-9.eE+-]+) s\), ncclCommAbort ([0-9.]+) s, "
                       r"stream drain ([0-9.]+) s", m.rccl_status)
        assert ph, m.rccl_status
        poll, limit, abort, drain = (float(v) for v in ph.groups())
        print(f"stalled group: sweep + recovery {dt:.3f} s = poll {poll:.3f} s (limit {limit} s) + ncclCommAbort "
              f"{abort:.3f} s + drain {drain:.3f} s + the sweep and the host copies")
        assert limit == 0.4 and 0.4 <= poll < 0.6, (poll, m.rccl_status)   # bounded by the limit
        assert drain < 0.2, (drain, m.rccl_status)  # the flag released the kernel: the stream was free at once
        assert dt - abort < 1.0, (dt, abort)        # everything that is this library's
        assert dt < 1.5, (dt, m.rccl_status)        # measured: 0.91 s = 0.40 poll + 0.51 ncclCommAbort (RCCL 2.26.6)
        monkeypatch.delenv("COREG_RCCL_TEST_STALL")
        again = _multi_carr(m, small, hs, large, hl, lags)  # RCCL is gone for this handle: host copies
        assert np.array_equal(again, want, equal_nan=True) and m.collective == "host-copy"
    # a bootstrap that hangs: the handle comes up without RCCL after the limit and still sweeps correctly
    monkeypatch.setenv("COREG_RCCL_TEST_INIT_STALL_SECONDS", "1.5")
    monkeypatch.setenv("COREG_RCCL_SELFTEST_SECONDS", "0.3")
    t0 = time.perf_counter()
    with _lib.MultiHandle(device_ids=[0]) as m:
        assert time.perf_counter() - t0 < 1.4
        assert m.rccl_status == "ncclCommInitAll did not return in time" and m.collective != "rccl"
        got = _multi_carr(m, small, hs, large, hl, lags)
        assert np.array_equal(got, want, equal_nan=True)
    time.sleep(1.5)  # (let the helper thread finish its bootstrap before the environment changes under it)


def test_alignment_uses_every_visible_gpu_from_a_plain_script(tmp_path, monkeypatch):
    """`Alignment(..., parallelism=True).align_using_carrington()` with no torch.distributed: the library drives every
    (here: virtual) GPU itself; same map as the single-device context, and an explicit device= keeps the latter."""
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.hdrshift import Alignment
    small, hs, large, hl, truth = H.scene()
    lag1, lag2 = 17.0 + 1.0 * (np.arange(20) - 10), -9.0 + 1.0 * (np.arange(20) - 10)

    def run(**kw):
        A = Alignment((large, hl), (small, hs), lag_crval1=lag1, lag_crval2=lag2, lag_cdelt1=None, lag_cdelt2=None,
                      lag_crota=[0.0], parallelism=True, **kw)
        corr = A.align_using_carrington(lonlims=H.CARR_LON, latlims=H.CARR_LAT, shape=SHAPE, return_type="corr")
        return corr, A

    single, A1 = run(device=0)
    assert "n_devices" not in A1.last_stats
    monkeypatch.setenv("COREG_VIRTUAL_DEVICES", "2")
    _lib._close_shared()
    try:
        multi, A2 = run()
        assert A2.last_stats["n_devices"] == 2 and A2.last_sharding == "blocks"
        assert np.nanmax(np.abs(multi - single)) <= 1e-12 and np.nanargmax(multi) == np.nanargmax(single)
        # a 3-D sweep: the CROTA combinations are dealt to the devices
        A3 = Alignment((large, hl), (small, hs), lag_crval1=lag1, lag_crval2=lag2, lag_cdelt1=None, lag_cdelt2=None,
                       lag_crota=[0.0, 0.3], parallelism=True)
        c3 = A3.align_using_carrington(lonlims=H.CARR_LON, latlims=H.CARR_LAT, shape=SHAPE, return_type="corr")
        assert A3.last_sharding == "combos" and np.nanmax(np.abs(c3[:, :, 0, 0, 0] - single[:, :, 0, 0, 0])) <= 1e-12
        # helioprojective frame, FITS in, through the same driver
        B = Alignment((large, hl), (small, hs), lag_crval1=lag1, lag_crval2=lag2, lag_cdelt1=None, lag_cdelt2=None,
                      lag_crota=None, parallelism=True)
        helio = B.align_using_helioprojective(return_type="corr")
        assert B.last_stats["n_devices"] == 2
        want = H.oracle_helio(small, hs, large, hl, (lag1[8:12], lag2[8:12], None, None, None))
        assert np.nanmax(np.abs(helio[8:12, 8:12] - want)) <= 1e-7
    finally:
        _lib._close_shared()


def test_bench_launch_threads_two_virtual_devices():
    """`python bench.py --gpus 2 --launch threads`: one process, the library's own multi-GPU driver (two logical devices
    on the one GPU of the box): the headline map with the injected shift as argmax, equal to device 0 sweeping alone."""
    from tests.test_api_cpu import _run_bench
    out = _run_bench({"COREG_VIRTUAL_DEVICES": "2"}, "--gpus", "2", "--launch", "threads", "--steps", "3", "--warmup",
                     "1", timeout=900)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["value"] > 0
    assert out["config"]["launch"] == "threads" and out["config"]["lag_sharding"] == "blocks"
    assert out["argmax_lag_arcsec"] == out["injected_shift_arcsec"][:2] == [17.0, -9.0]
    assert out["map_vs_single_gpu"]["max_abs_diff"] <= 1e-12 and out["map_vs_single_gpu"]["same_argmax"]
    assert len(out["per_rank"]) == 2 and all(r["kernel_ms"] > 0 for r in out["per_rank"])
    assert out["pcie_inclusive"]["identical_to_resident_map"]


def test_multi_rccl_in_a_script_that_never_imports_torch(tmp_path):
    """The plain user script of the reference's README imports no torch: the library then loads PyTorch's HIP runtime
    itself (two HIP runtimes in one process do not both see the GPU) and the multi-GPU driver dlopens the RCCL that goes
    with it.  One-rank group on the one GPU; the map equals the single-device one."""
    import os
    import subprocess
    import sys
    from tests.test_api_cpu import ROOT
    script = tmp_path / "plain.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from euispice_coreg_amd import _lib\n"
        "from tests import helpers as H\n"
        "assert 'torch' not in sys.modules\n"
        "small, hs, large, hl, _ = H.scene()\n"
        "grid = _lib.Grid(H.CARR_LON, H.CARR_LAT, (72, 64))\n"
        "lags = _lib.LagSet(17.0 + np.arange(-6, 6.0), -9.0 + np.arange(-6, 6.0), None, None, None)\n"
        "with _lib.MultiHandle(device_ids=[0]) as m:\n"
        "    m.set_small(small); m.prepare_reference_carrington(large, hl, grid, 1.004, 2)\n"
        "    got = m.sweep_carrington(hs, grid, 1.004, lags)\n"
        "    print('collective', m.collective, 'mode', m.last_mode)\n"
        "with _lib.CoregHandle(0) as h:\n"
        "    h.set_small(small); h.prepare_reference_carrington(large, hl, grid, 1.004, 2)\n"
        "    want = h.sweep_carrington(hs, grid, 1.004, lags)\n"
        "assert np.array_equal(got, want, equal_nan=True)\n"
        "assert 'torch' not in sys.modules\n"
        "print('ok')\n")
    env = dict(os.environ, COREG_MULTI_FORCE_RCCL="1")
    env.pop("COREG_VIRTUAL_DEVICES", None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "ok" in r.stdout
    assert "collective rccl mode slices" in r.stdout, r.stdout


def test_multi_on_two_or_more_physical_gpus(gpu_handle, monkeypatch):
    """The real thing, wherever more than one GPU is visible (the build's box has one: skipped there): every device from
    one process, RCCL communicators from ncclCommInitAll, the image by row shares + one all-gather, every partition of
    the lag set with its one collective -- against device 0 sweeping alone."""
    from euispice_coreg_amd import _lib, parallel
    monkeypatch.delenv("COREG_VIRTUAL_DEVICES", raising=False)
    n = _lib.physical_device_count()
    if n < 2:
        pytest.skip("one GPU visible")
    small, hs, large, hl, _ = H.scene()
    l24, l23 = 17.0 + 1.0 * (np.arange(24) - 12), -9.0 + 1.0 * (np.arange(23) - 11)
    cases = [(l24, l23, None, None, [0.3]), (l24, l23, None, None, [0.0, 0.3, -0.2]),
             (17.0 + 2.0 * (np.arange(5) - 2), -9.0 + 2.0 * (np.arange(4) - 2), None, None, None)]
    with _lib.MultiHandle() as m:
        assert m.size == n
        print("collective:", m.collective, "rccl:", m.rccl_status)
        for lags in cases:
            for force in (-1, 1, 2, 4):
                m.set_option("force_mode", force)
                got = _multi_carr(m, small.astype(np.float32), hs, large, hl, lags)
                want = _single_carr(gpu_handle, small.astype(np.float32), hs, large, hl, lags)
                assert np.array_equal(np.isnan(got), np.isnan(want)), (force, m.last_mode)
                assert np.nanmax(np.abs(got - want)) <= 1e-12, (force, m.last_mode)
        m.set_option("force_mode", -1)
        assert m.collective in ("rccl", "host-copy")


def test_multi_helioprojective_and_plate_carree_3d_lag_sets(gpu_handle, monkeypatch):
    """3-D lag sets (CROTA values) in the other two frames over three logical devices: the planner's choice (a
    helioprojective sweep is ONE launch whatever the lag set, a plate-carree sweep one per combination -- the cost model
    knows) and every forced partition give the single-device map."""
    from euispice_coreg_amd import _lib, parallel, synthetic
    monkeypatch.setenv("COREG_VIRTUAL_DEVICES", "3")
    small, hs, large, hl, _ = H.scene()
    lags = (np.arange(-8.0, 20.0, 1.0), np.arange(-14.0, 3.0, 1.0), None, None, [0.0, 0.3, -0.2])
    ls = _lib.LagSet(*lags)
    want = H.gpu_helio(gpu_handle, small, hs, large, hl, lags).ravel()
    auto = parallel.lag_sharding(ls.shape, 3, per_combo_launch=False)
    assert auto in ("blocks", "combos")
    with _lib.MultiHandle() as m:
        m.set_small(small)
        m.prepare_reference_helioprojective(large, hl, hs, 2)
        for force, mode in ((-1, auto), (1, "blocks"), (4, "combos"), (2, "slices")):
            m.set_option("force_mode", force)
            got = m.sweep_helioprojective(hs, hs, ls)
            assert m.last_mode == mode
            assert np.array_equal(np.isnan(got), np.isnan(want)) and np.nanmax(np.abs(got - want)) <= 1e-12, mode
        m.set_option("force_mode", -1)
        # plate-carree maps (align_using_initial_carrington): 40 x 40 lags x 3 CROTA values
        cs, chs, cl, chl, _ = synthetic.make_car_scene(small_shape=(60, 70), large_shape=(90, 100))
        clags = (np.linspace(-0.02, 0.02, 40), np.linspace(-0.015, 0.015, 40), None, None, [0.0, 0.2, -0.1])
        cls = _lib.LagSet(*clags)
        gpu_handle.set_small(cs)
        gpu_handle.set_reference_on_grid(np.asarray(cl, dtype=np.float32))
        wantc = gpu_handle.sweep_helioprojective(chl, chs, cls)
        m.set_small(cs)
        m.set_reference_on_grid(np.asarray(cl, dtype=np.float32))
        gotc = m.sweep_helioprojective(chl, chs, cls)
        assert m.last_mode == "combos"
        assert np.array_equal(np.isnan(gotc), np.isnan(wantc)) and np.nanmax(np.abs(gotc - wantc)) <= 1e-12


def test_multi_odd_order_noise_decided_samples(gpu_handle, monkeypatch):
    """An odd spline order under an unrotated header, lag axes through 0 (pure CRVAL1 and pure CRVAL2 lags bring curves of
    coordinates back onto integers; the zero lag, its whole grid): every device re-evaluates the noise-decided samples of
    its own lag-points with wcslib's chain ("tap_fix"), and -- in the grid-share mode -- device 0 carries the
    correction for all.  Same map as the single handle, and the oracle's."""
    from euispice_coreg_amd import _lib
    monkeypatch.setenv("COREG_VIRTUAL_DEVICES", "3")
    small, hs, large, hl, _ = H.scene(nan_frac=0.02)
    hs = dict(hs, CROTA=0.0, PC1_1=1.0, PC1_2=0.0, PC2_1=0.0, PC2_2=1.0)
    for order, lags in ((1, (np.arange(-6.0, 7.0, 3.0), np.arange(-4.0, 5.0, 2.0), None, None, [0.0, 0.3])),
                        (3, (np.array([0.0, 2.5]), np.array([0.0]), None, None, None))):  # (2 lag-points: grid shares)
        want = H.oracle_helio(small, hs, large, hl, lags, order=order)
        single = H.gpu_helio(gpu_handle, small, hs, large, hl, lags, order=order)
        assert gpu_handle.last_tap_fix()["samples"] > 0
        H.assert_corr_close(single, want, 1e-7, f"single handle, order {order}")
        ls = _lib.LagSet(*lags)
        with _lib.MultiHandle() as m:
            m.set_small(small)
            m.prepare_reference_helioprojective(large, hl, hs, order)
            modes = set()
            for force in ((-1, 1, 4, 2) if ls.size > 2 else (-1,)):
                m.set_option("force_mode", force)
                got = m.sweep_helioprojective(hs, hs, ls, order=order).reshape(single.shape)
                modes.add(m.last_mode)
                assert np.array_equal(np.isnan(got), np.isnan(single))
                assert np.nanmax(np.abs(got - single)) <= 1e-12, (order, m.last_mode)
                H.assert_corr_close(got, want, 1e-7, f"{m.last_mode}, order {order}")
            m.set_option("force_mode", -1)
            assert "points" in modes  # (the planner's choice for lag sets this small: device 0 carries the correction)
            assert ls.size <= 2 or {"blocks", "combos", "slices"} <= modes


def test_multi_tile_compressed_images_on_every_device(gpu_handle, monkeypatch, tmp_path):
    """Tile-compressed files (EUI's format) through the all-GPU driver: every device uploads the compressed bytes -- a
    fraction of the pixels' -- and decodes them itself, for the image to align and for the reference image, both
    frames.  Same maps as the single handle on the host-decoded pixels."""
    from euispice_coreg_amd import _lib
    from euispice_coreg_amd.utils import fits_io
    monkeypatch.setenv("COREG_VIRTUAL_DEVICES", "3")
    small, hs, large, hl, _ = H.scene()
    ps, pl = str(tmp_path / "s.fits"), str(tmp_path / "l.fits")
    fits_io.write_compressed_image(ps, small.astype(np.float32), hs, quantize="SUBTRACTIVE_DITHER_2", dither0=77, tile=(32, 8))
    fits_io.write_compressed_image(pl, np.clip(np.nan_to_num(large) * 8.0, 0, 65535).astype(np.uint16), hl)
    cs, cl = fits_io.open_compressed(ps, -1), fits_io.open_compressed(pl, -1)
    assert cs.on_gpu and cl.on_gpu
    ds, dl = fits_io.native_pixels(cs.decode()), fits_io.native_pixels(cl.decode())
    lags = (np.arange(9.0, 25.0, 2.0), np.arange(-15.0, -3.0, 2.0), None, None, [0.0, 0.3])
    want_c = _single_carr(gpu_handle, ds, hs, dl, hl, lags)
    want_h = H.gpu_helio(gpu_handle, ds, hs, dl, hl, lags).ravel()
    with _lib.MultiHandle() as m:
        got_c = _multi_carr(m, cs, hs, cl, hl, lags)
        m.set_small(cs)
        m.prepare_reference_helioprojective(cl, hl, hs, 2)
        got_h = m.sweep_helioprojective(hs, hs, _lib.LagSet(*lags))
    assert np.array_equal(np.isnan(got_c), np.isnan(want_c)) and np.nanmax(np.abs(got_c - want_c)) <= 1e-12
    assert np.array_equal(np.isnan(got_h), np.isnan(want_h)) and np.nanmax(np.abs(got_h - want_h)) <= 1e-12
    assert np.isfinite(got_c).any() and np.isfinite(got_h).any()
