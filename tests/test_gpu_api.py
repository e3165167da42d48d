"""GPU tests of the drop-in API: euispice_coreg_amd.hdrshift.Alignment used exactly like the reference's
(README.md:47-87, 97-139) on synthetic FITS files, checked against the oracle's sweep."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fits_pair(tmp_path_factory):
    from euispice_coreg_amd.utils import fits_io
    d = tmp_path_factory.mktemp("fits")
    small, hs, large, hl, truth = H.scene(small_n=128, large_n=192, seed=11, n_blobs=200)
    ps, pl = str(d / "small.fits"), str(d / "large.fits")
    fits_io.write_images(ps, [(None, {}), (small.astype(np.float32), hs)])
    fits_io.write_images(pl, [(None, {}), (large.astype(np.float32), hl)])
    return ps, pl, small, hs, large, hl, truth


def test_align_using_carrington_dropin(fits_pair):
    from euispice_coreg_amd.hdrshift import Alignment, AlignmentResults
    ps, pl, small, hs, large, hl, truth = fits_pair
    lag1, lag2 = np.arange(5, 30, 4.0), np.arange(-21, 4, 4.0)
    kw = dict(lonlims=(228, 262), latlims=(-12, 22), shape=(96, 96))
    A = Alignment(large_fov_known_pointing=pl, small_fov_to_correct=ps, lag_crval1=lag1, lag_crval2=lag2,
                  lag_cdelt1=[0], lag_cdelt2=[0], lag_crota=[0.0, 0.3], parallelism=True, counts_cpu_max=20,
                  small_fov_value_max=2500.0)
    res = A.align_using_carrington(method="correlation", **kw)
    assert isinstance(res, AlignmentResults) and res.corr.shape == (7, 7, 1, 1, 2, 1)
    sm = small.astype(np.float32).astype(np.float64)
    from oracle import coreg_oracle as O
    O.set_threshold_minmax_to_nan(sm, None, 2500.0)
    want = H.oracle_carrington(sm, hs, large.astype(np.float32).astype(np.float64), hl,
                               (lag1, lag2, [0], [0], [0.0, 0.3]), kw["shape"], kw["lonlims"], kw["latlims"])
    H.assert_corr_close(res.corr, want, 1e-10, "Alignment.align_using_carrington")
    assert tuple(res.max_index) == tuple(np.unravel_index(np.nanargmax(want), want.shape))
    # raw array form
    A2 = Alignment(pl, ps, lag1, lag2, [0], [0], [0.0, 0.3], parallelism=True, small_fov_value_max=2500.0)
    corr = A2.align_using_carrington(return_type="corr", **kw)
    assert np.array_equal(corr, res.corr, equal_nan=True)


@pytest.mark.parametrize("parallelism", [True, False])
def test_align_using_helioprojective_dropin(fits_pair, parallelism):
    from euispice_coreg_amd.hdrshift import Alignment
    ps, pl, small, hs, large, hl, truth = fits_pair
    lag1, lag2 = np.arange(9, 26, 4.0), np.arange(-17, 0, 4.0)
    A = Alignment(pl, ps, lag_crval1=lag1, lag_crval2=lag2, lag_cdelt1=None, lag_cdelt2=None, lag_crota=[0.3],
                  parallelism=parallelism, small_fov_value_min=20.0)
    res = A.align_using_helioprojective(method="correlation")
    sm = small.astype(np.float32).astype(np.float64)
    from oracle import coreg_oracle as O
    O.set_threshold_minmax_to_nan(sm, 20.0, None)
    want = H.oracle_helio(sm, hs, large.astype(np.float32).astype(np.float64), hl, (lag1, lag2, None, None, [0.3]),
                          parallelism=parallelism)
    H.assert_corr_close(res.corr, want, 1e-7, f"Alignment.align_using_helioprojective parallelism={parallelism}")
    assert len(res.shift_arcsec) == 5 and res.unit_lag == "arcsec"


def test_fov_limits_and_remove_fov_limits(fits_pair):
    """alignment.py:844-874, 1082-1127: box set to NaN, then re-grid of the small image on a regular sub-FOV grid
    (GPU resample), then the sweep -- against the oracle's restatement of the same pre-processing."""
    from euispice_coreg_amd.hdrshift import Alignment
    from oracle import coreg_oracle as O
    ps, pl, small, hs, large, hl, truth = fits_pair
    lag1, lag2 = np.arange(9, 26, 4.0), np.arange(-17, 0, 4.0)
    fov = [[-700.0, -100.0], [100.0, 700.0]]      # arcsec: a square part of the small FOV
    rem = [[-450.0, -400.0], [350.0, 420.0]]      # arcsec: box to blank
    A = Alignment(pl, ps, lag_crval1=lag1, lag_crval2=lag2, lag_cdelt1=None, lag_cdelt2=None, lag_crota=None,
                  parallelism=True)
    res = A.align_using_helioprojective(fov_limits=fov, remove_fov_limits=rem)
    sm = small.astype(np.float32).astype(np.float64)
    st = H.oracle_state(sm, hs, large.astype(np.float32).astype(np.float64), hl, (lag1, lag2, None, None, None))
    O.set_remove_fov_limits_to_nan(st, [v / 3600.0 for v in rem[0]], [v / 3600.0 for v in rem[1]])
    assert np.isnan(st.data_small).sum() > np.isnan(sm).sum()
    O.select_fov_in_small_data(st, [v / 3600.0 for v in fov[0]], [v / 3600.0 for v in fov[1]])
    assert st.data_small.shape == A.data_small.shape and st.data_small.shape[0] < small.shape[0]
    m = np.isfinite(st.data_small)
    assert np.array_equal(np.isfinite(A.data_small), m)
    assert np.abs(A.data_small[m] - st.data_small[m]).max() <= 1e-9 * np.nanmax(sm)
    want = O.find_best_header_parameters(st, "helioprojective")
    H.assert_corr_close(res.corr, want, 1e-7, "fov_limits + remove_fov_limits")


def test_alignment_refuses_unimplemented_paths(fits_pair):
    from euispice_coreg_amd.hdrshift import Alignment
    ps, pl = fits_pair[0], fits_pair[1]
    A = Alignment(pl, ps, [0.0], [0.0], None, None, None)
    with pytest.raises(NotImplementedError):
        A.align_using_helioprojective(method="no-such-method")
    with pytest.raises(NotImplementedError):
        A.align_using_carrington(lonlims=(228, 262), latlims=(-12, 22), shape=(32, 32),
                                 method_carrington_reprojection="sunpy")


def test_alignment_spice_l2_dropin():
    """AlignmentSpice on a synthetic L2 raster cube (SURVEY cfg 4 shape: CDELT1 != CDELT2, 2-D header in degrees after
    flattening, lags in arcsec) against the oracle run on the same prepared image / header."""
    from euispice_coreg_amd import synthetic
    from euispice_coreg_amd.hdrshift import AlignmentSpice, AlignmentResults
    cube, h4, large, hl, truth = synthetic.make_spice_l2()
    lag1, lag2 = np.arange(-31.0, -14.0, 2.0), np.arange(28.0, 45.0, 2.0)
    wave_lo, wave_hi = 976.6, 977.5
    A = AlignmentSpice((large, hl), (cube, h4), lag_crval1=lag1, lag_crval2=lag2, lag_crota=np.array([0.0]),
                       lag_cdelt1=np.array([0.0]), lag_cdelt2=np.array([0.0]), parallelism=True, level=2,
                       wavelength_interval_to_sum=[wave_lo, wave_hi])
    with pytest.warns(UserWarning):  # "Units of headers in deg: modifying inputs units"
        res = A.align_using_helioprojective(method="correlation")
    assert isinstance(res, AlignmentResults) and res.corr.shape == (9, 9, 1, 1, 1, 1) and res.unit_lag == "arcsec"
    want = H.oracle_helio(A.data_small, A.hdr_small, large, hl, (lag1, lag2, [0.0], [0.0], [0.0]), parallelism=True,
                          unit_lag="arcsec")
    H.assert_corr_close(res.corr, want, 1e-7, "AlignmentSpice.align_using_helioprojective")
    am = res.max_index
    assert (lag1[am[0]], lag2[am[1]]) == (truth["lag_crval1"], truth["lag_crval2"])
    assert abs(res.shift_arcsec[0] - truth["lag_crval1"]) < 1.0 and abs(res.shift_arcsec[1] - truth["lag_crval2"]) < 1.0
    # Carrington frame on the same raster (2-D header converted to arcsec, alignment_spice.py:159-168)
    B = AlignmentSpice((large, hl), (cube, h4), lag_crval1=lag1, lag_crval2=lag2, parallelism=True, level=2)
    kw = dict(lonlims=(241.0, 247.0), latlims=(3.5, 9.0), shape=(60, 110))
    resc = B.align_using_carrington(**kw)
    assert B.hdr_small["CUNIT1"] == "arcsec"
    wantc = H.oracle_carrington(B.data_small, B.hdr_small, large, hl, (lag1, lag2, None, None, None), kw["shape"],
                                kw["lonlims"], kw["latlims"])
    H.assert_corr_close(resc.corr, wantc, 1e-10, "AlignmentSpice.align_using_carrington")


def test_bench_two_ranks_on_one_gpu_gloo():
    """bench.py --gpus 2 started as a plain command on the one-GPU box (ranks share the device, all-gather over gloo):
    real sweeps on both ranks, the assembled map has the injected shift as its argmax, per-rank kernel times present."""
    from tests.test_api_cpu import _run_bench
    out = _run_bench({"COREG_BENCH_BACKEND": "gloo", "COREG_CPU_CORES": "8"}, "--gpus", "2", "--steps", "4",
                     "--warmup", "2", timeout=900)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["value"] > 0
    assert out["argmax_lag_arcsec"] == out["injected_shift_arcsec"][:2] == [17.0, -9.0]
    assert len(out["per_rank"]) == 2 and all(r["kernel_ms"] > 0 and r["lags"] == 1800 for r in out["per_rank"])
    assert 0.0 < out["roofline"]["frac"] <= 1.0
    assert out["config"]["lag_sharding"] == "blocks"  # the partition hdrshift.Alignment uses for this lag set
    # the N > 1 line verifies itself: the assembled map against one GPU sweeping everything, against the oracle on a
    # CPU sample, and the BASELINE.md section 2 call (host images in, host map out) with every rank taking part
    m = out["map_vs_single_gpu"]
    assert m["max_abs_diff"] <= 1e-12 and m["same_argmax"] and m["same_nan_pattern"]
    p = out["parity_vs_cpu_sample"]
    assert p["n"] == 128 and p["max_abs_dcorr"] <= 1e-10 and p["argmax_on_sample_equal"]
    c = out["pcie_inclusive"]
    assert c["n_gpus"] == 2 and c["value"] > 0 and c["ms_per_step"] > 0 and c["identical_to_resident_map"]
    assert out["cpu_baseline"] is None  # the CPU leg is timed at N = 1 only


def test_bench_two_ranks_point_sharded_gloo():
    """The same with the grid sharded instead of the lag plane: every rank sweeps all 3600 lag-points over its share of
    the grid's points, one all-reduce of the six sums per lag."""
    from tests.test_api_cpu import _run_bench
    out = _run_bench({"COREG_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "4", "--warmup", "2",
                     "--no-cpu-baseline", "--shard", "points", timeout=900)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["value"] > 0
    assert "all-reduce" in out["config"]["parallelism"]
    assert out["argmax_lag_arcsec"] == out["injected_shift_arcsec"][:2] == [17.0, -9.0]
    assert all(r["kernel_ms"] > 0 and r["lags"] == 3600 for r in out["per_rank"])
    m = out["map_vs_single_gpu"]
    assert m["max_abs_diff"] <= 1e-12 and m["same_argmax"] and m["same_nan_pattern"]
    assert out["pcie_inclusive"]["n_gpus"] == 2 and out["pcie_inclusive"]["value"] > 0


def test_alignment_two_ranks_gloo_every_sharding_mode(tmp_path):
    """`Alignment` under torch.distributed (two ranks sharing the one GPU, gloo): a 5 x 5 lag set is below the
    point-sharding threshold (grid shares + one all-reduce of the six sums per lag), a 24 x 24 one above it: with one
    CROTA value blocks of the lag plane + one all-gather (the partition bench.py times), with two or three the CROTA
    combinations are dealt to the ranks (each sweeps the whole plane); all must give the single-process map on every
    rank."""
    import os
    import subprocess
    import sys
    from tests.test_api_cpu import ROOT
    script = tmp_path / "worker.py"
    script.write_text(
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch.distributed as dist\n"
        "from euispice_coreg_amd import parallel\n"
        "from euispice_coreg_amd.hdrshift import Alignment\n"
        "from tests import helpers as H\n"
        "small, hs, large, hl, _ = H.scene()\n"
        "def run(l1, l2, crota=(0.0, 0.3)):\n"
        "    A = Alignment((large, hl), (small, hs), lag_crval1=l1, lag_crval2=l2, lag_cdelt1=None, lag_cdelt2=None,\n"
        "                  lag_crota=list(crota), parallelism=True)\n"
        "    return A.align_using_carrington(lonlims=H.CARR_LON, latlims=H.CARR_LAT, shape=(72, 64), return_type='corr')\n"
        "small_set = (17.0 + 2.0 * (np.arange(5) - 2), -9.0 + 2.0 * (np.arange(5) - 2))\n"
        "big_set = (17.0 + 1.0 * (np.arange(24) - 12), -9.0 + 1.0 * (np.arange(24) - 12))\n"
        "single = [run(*small_set), run(*big_set), run(*big_set, crota=(0.3,)), run(*big_set, crota=(0.0, 0.3, -0.2))]\n"
        "dist.init_process_group('gloo')\n"
        "rank, world = parallel.world_info()\n"
        "assert parallel.lag_sharding((5, 5, 1, 1, 2), world) == 'points'\n"
        "assert parallel.lag_plan((24, 24, 1, 1, 2), world) == ('combos', 2, 1, 1)\n"
        "assert parallel.lag_sharding((24, 24, 1, 1, 1), world) == 'blocks'\n"
        "assert parallel.lag_plan((24, 24, 1, 1, 3), world)[:2] == ('combos', 2)   # runs of 2 and 1 combinations\n"
        "multi = [run(*small_set), run(*big_set), run(*big_set, crota=(0.3,)), run(*big_set, crota=(0.0, 0.3, -0.2))]\n"
        "for a, b in zip(single, multi):\n"
        "    assert a.shape == b.shape and np.nanmax(np.abs(a - b)) <= 1e-12, np.nanmax(np.abs(a - b))\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "print('rank', rank, 'ok')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29571", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


def test_readme_example_of_the_reference_runs_to_its_last_line(fits_pair, tmp_path):
    """The reference's README example (README.md:47-87), statement for statement with the package name changed: construct,
    align, write the corrected file, plot the correlation, plot the co-alignment.  The last figure is drawn from the
    image to align resampled onto the reference image's grid by the library, before and after the correction: the
    corrected version must follow the reference image better than the uncorrected one."""
    import os
    pytest.importorskip("matplotlib")
    from euispice_coreg_amd.hdrshift import Alignment
    ps, pl, small, hs, large, hl, truth = fits_pair
    param_alignment = {"lag_crval1": np.arange(-30, 30, 2), "lag_crval2": np.arange(-30, 30, 2), "lag_crota": np.array([0]),
                       "lag_cdelt1": np.array([0]), "lag_cdelt2": np.array([0])}
    windows = [-1]
    A = Alignment(large_fov_known_pointing=pl, small_fov_to_correct=ps, parallelism=True, display_progress_bar=True,
                  counts_cpu_max=20, **param_alignment)
    results = A.align_using_helioprojective(method="correlation")
    path_save_fits = str(tmp_path / "fits.fits")
    results.write_corrected_fits(windows, path_to_l3_output=path_save_fits)
    results.plot_correlation(path_save_figure=os.path.join(str(tmp_path), "correlation_results.pdf"))
    results.plot_co_alignment(path_save_figure=os.path.join(str(tmp_path), "co_alignment_results.pdf"))
    for name in ("fits.fits", "correlation_results.pdf", "co_alignment_results.pdf"):
        assert os.path.getsize(os.path.join(str(tmp_path), name)) > 1000
    assert abs(results.shift_arcsec[0] - truth["lag_crval1"]) < 3.0 and abs(results.shift_arcsec[1] - truth["lag_crval2"]) < 3.0  # (7.9 arcsec pixels, a roll error outside the lag set)
    c = results.co_alignment

    def pearson(a, b):
        m = np.isfinite(a) & np.isfinite(b)
        return np.corrcoef(a[m], b[m])[0, 1]
    assert pearson(c["reference"], c["after"]) > pearson(c["reference"], c["before"]) + 0.05
    assert pearson(c["reference"], c["after"]) > 0.5
    # limits of the shown part in the lag unit; SPICE cubes and the other plot types are refused, not mis-drawn
    results.plot_co_alignment(lonlims=(-600, -100), latlims=(200, 600))
    with pytest.raises(NotImplementedError):
        results.plot_co_alignment(type_plot="sunpy")


def test_readme_spice_example_runs_to_its_last_line(tmp_path):
    """The reference's SPICE example (README.md:172-212) on a synthetic level-2 raster file: AlignmentSpice from paths, a
    window chosen by name, write_corrected_fits on a list of window names, plot_correlation, and plot_co_alignment with
    two contour levels -- drawn from the spectrally summed, slit-masked image the sweep saw and its flattened header."""
    import os
    pytest.importorskip("matplotlib")
    from euispice_coreg_amd import synthetic
    from euispice_coreg_amd.hdrshift import AlignmentSpice
    from euispice_coreg_amd.utils import fits_io
    cube, h4, large, hl, truth = synthetic.make_spice_l2()
    name = "Ly-gamma-CIII group (Merged)"
    path_spice = str(tmp_path / "solo_L2_spice-n-ras_20220317T000032_V01.fits")
    path_sr = str(tmp_path / "synthetic_raster.fits")
    fits_io.write_images(path_spice, [(cube.astype(np.float32), dict(h4, EXTNAME=name))])
    fits_io.write_images(path_sr, [(None, {}), (large.astype(np.float32), hl)])
    param_alignment = {"lag_crval1": np.arange(-31.0, -14.0, 2.0), "lag_crval2": np.arange(28.0, 45.0, 2.0),
                       "lag_crota": np.array([0]), "lag_cdelt1": np.array([0]), "lag_cdelt2": np.array([0])}
    A = AlignmentSpice(large_fov_known_pointing=path_sr, small_fov_to_correct=path_spice, display_progress_bar=True,
                       parallelism=True, counts_cpu_max=10, large_fov_window=-1, small_fov_window=name,
                       path_save_figure=str(tmp_path), **param_alignment)
    with pytest.warns(UserWarning):
        results = A.align_using_helioprojective(method="correlation")
    am = results.max_index
    assert (param_alignment["lag_crval1"][am[0]], param_alignment["lag_crval2"][am[1]]) == (truth["lag_crval1"], truth["lag_crval2"])
    out = str(tmp_path / "corrected.fits")
    results.write_corrected_fits([name], path_to_l3_output=out)
    h_out = fits_io.read_header(out, name)
    assert h_out["NAXIS"] == 4 and h_out["CRVAL1"] != h4["CRVAL1"]
    results.plot_correlation(path_save_figure=os.path.join(str(tmp_path), "correlation_results.pdf"))
    results.plot_co_alignment(path_save_figure=os.path.join(str(tmp_path), "co_alignment_results.pdf"), levels_percentile=[80, 90])
    assert os.path.getsize(os.path.join(str(tmp_path), "co_alignment_results.pdf")) > 1000
    c = results.co_alignment

    def pearson(a, b):
        m = np.isfinite(a) & np.isfinite(b)
        return np.corrcoef(a[m], b[m])[0, 1]
    assert pearson(c["reference"], c["after"]) > pearson(c["reference"], c["before"])


@pytest.mark.parametrize("name", ["order1", "cfg4"])
def test_bench_config_lines_carry_the_variant_s_roofline(name):
    """`python bench.py --config NAME`: one JSON line with the headline's schema for another sweep variant --
    the variant's own counted flop, ps per (point x lag), kernel time from the library's HIP events, and a CPU sample of
    that config checked against the map."""
    from tests.test_api_cpu import _run_bench
    out = _run_bench({"COREG_CPU_CORES": "8"}, "--config", name, "--steps", "2", "--warmup", "1", "--cpu-sample", "8",
                     timeout=600)
    r = out["roofline"]
    assert out["config"]["name"] == name and out["n_gpus"] == 1 and out["value"] > 0 and out["unit"] == "lag-points/s"
    assert r["flop_per_point_lag"] == {"order1": 23.0, "cfg4": 62.0}[name] and 0.0 < r["frac"] < 1.0
    assert r["kernel"].startswith({"order1": "k_sweep<TRANSLATE,1", "cfg4": "k_sweep<HOMOGRAPHY,2"}[name])
    assert 0.5 < r["ps_per_point_lag"] < 5.0 and r["kernel_ms"] > 0 and r["lags_per_launch"] * r["launches_per_sweep"] == out["config"]["lag_points"]
    p = out["parity_vs_cpu_sample"]
    assert p["n"] == 8 and p["max_abs_dcorr"] <= p["tolerance"]
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
