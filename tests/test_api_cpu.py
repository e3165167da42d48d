"""CPU tests of the host-side mirror of the reference interface: AlignmentResults (pinned by the reference's own
fixture), the minimal FITS reader/writer, header correction, and the multi-rank lag sharding (gloo, world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.test_oracle_golden import REF_CORR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_alignment_results_reference_fixture():
    """euispice_coreg/hdrshift/test/test_AlignmentResults.py:161-173 (tolerance 2e-2)."""
    from euispice_coreg_amd.hdrshift import AlignmentResults
    R = AlignmentResults(corr=REF_CORR, lag_crval1=np.arange(15, 26, 1), lag_crval2=np.arange(5, 11, 1),
                         lag_cdelt1=None, lag_cdelt2=[0], lag_crota=[0.75], unit_lag="arcsec")
    assert tuple(int(i) for i in R.max_index) == (9, 1, 0, 0, 0, 0)
    assert abs(R.shift_pixels[0] - 9.33682107) < 2e-2
    assert abs(R.shift_pixels[1] - 1.42187891) < 2e-2
    assert abs(R.shift_arcsec[0] - (15 + R.shift_pixels[0])) < 1e-9
    assert R.shift_arcsec[4] == 0.75
    assert "Shift" in str(R)
    # agrees with the oracle's restatement of the same routine (which calls scipy): the literal scipy call to the last
    # digit, the library's restatement of it (the default, csrc/fit.hpp) to 1e-6 px on this well-conditioned peak
    from oracle import coreg_oracle as O
    _, px, _ = O.compute_shift(REF_CORR, np.arange(15, 26, 1.0), np.arange(5, 11, 1.0))
    assert R.fit == "native" and R.fit_info["status"] > 0
    assert abs(px[0] - R.shift_pixels[0]) < 1e-6 and abs(px[1] - R.shift_pixels[1]) < 1e-6
    S = AlignmentResults(corr=REF_CORR, lag_crval1=np.arange(15, 26, 1), lag_crval2=np.arange(5, 11, 1),
                         lag_cdelt1=None, lag_cdelt2=[0], lag_crota=[0.75], unit_lag="arcsec", fit="scipy")
    assert abs(px[0] - S.shift_pixels[0]) < 1e-9 and abs(px[1] - S.shift_pixels[1]) < 1e-9


def test_alignment_results_too_few_points_falls_back_to_argmax():
    from euispice_coreg_amd.hdrshift import AlignmentResults
    corr = np.array([0.1, 0.3, 0.2]).reshape(3, 1, 1, 1, 1, 1)
    with pytest.warns(UserWarning):
        R = AlignmentResults(corr, [1.0, 2.0, 3.0], [0.0], None, None, None, "arcsec")
    assert R.shift_arcsec[0] == 2.0


def test_fits_roundtrip_and_corrected_fits(tmp_path):
    from euispice_coreg_amd.hdrshift import AlignmentResults
    from euispice_coreg_amd.utils import fits_io
    from tests import helpers as H
    small, hs, _, _, _ = H.scene(small_n=32, large_n=32)
    p = str(tmp_path / "small.fits")
    fits_io.write_images(p, [(None, {}), (small.astype(np.float32), hs)])
    data, hdr = fits_io.read_image(p, -1)
    assert data.dtype == np.float32 and np.array_equal(data, small.astype(np.float32), equal_nan=True)
    for k in ("CRVAL1", "CDELT2", "PC1_2", "CROTA", "DSUN_OBS", "NAXIS1"):
        assert hdr[k] == pytest.approx(hs[k], rel=1e-15), k
    assert hdr["CUNIT1"] == "arcsec" and hdr["DATE-AVG"] == hs["DATE-AVG"]
    R = AlignmentResults(REF_CORR, np.arange(15, 26, 1), np.arange(5, 11, 1), None, [0], [0.75], "arcsec",
                         image_to_align_path=p, image_to_align_window=-1)
    out = str(tmp_path / "corrected.fits")
    R.write_corrected_fits([-1], out)
    d2, h2 = fits_io.read_image(out, 1)
    assert np.array_equal(d2, data, equal_nan=True)
    assert h2["CRVAL1"] == pytest.approx(hs["CRVAL1"] + R.shift_arcsec[0], rel=1e-14)
    assert h2["CRVAL2"] == pytest.approx(hs["CRVAL2"] + R.shift_arcsec[1], rel=1e-14)
    assert h2["CROTA"] == pytest.approx(hs["CROTA"] + 0.75, rel=1e-14)
    assert h2["PC1_1"] == pytest.approx(np.cos(np.deg2rad(hs["CROTA"] + 0.75)), rel=1e-14)
    hc = R.return_corrected_header(-1)
    assert hc["CRVAL1"] == h2["CRVAL1"]
    with pytest.raises(ValueError):
        R.write_corrected_fits(["no-such-window"], out)


def test_alignment_constructor_and_errors():
    from euispice_coreg_amd.hdrshift import Alignment
    from tests import helpers as H
    small, hs, large, hl, _ = H.scene(small_n=32, large_n=32)
    A = Alignment((large, hl), (small, hs), lag_crval1=[0.0], lag_crval2=[0.0], lag_cdelt1=None, lag_cdelt2=None,
                  lag_crota=None)
    assert np.array_equal(A.lag_crota, [0.0]) and A.order == 2 and A.counts == 40
    with pytest.raises(ValueError):
        A.align_using_carrington(lonlims=(0, 1))  # "either set lonlims as None, or not. no in between."
    with pytest.raises(ValueError):
        A.align_using_carrington(lonlims=(0, 1), latlims=(0, 1), shape=(8, 8), method_carrington_reprojection="xx")
    hs2 = dict(hs)
    for k in ("PC1_1", "PC1_2", "PC2_1", "PC2_2", "CROTA"):
        hs2.pop(k)
    B = Alignment((large, hl), (small, hs2), [0.0], [0.0], None, None, None)
    with pytest.raises(ValueError):  # no CROTA / PCi_j and force_crota_0 not set (alignment.py:592)
        B.align_using_helioprojective()


def test_shard_bounds_cover_range():
    from euispice_coreg_amd import parallel
    for n in (0, 1, 7, 3600, 14641):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi, chunk = parallel.shard_bounds(n, world, r)
                assert 0 <= lo <= hi <= n and hi - lo <= chunk
                seen.extend(range(lo, hi))
            assert seen == list(range(n))


def test_block_sharding_covers_plane():
    """2-D block sharding of the (CRVAL1, CRVAL2) plane: blocks tile the plane, the gather index reassembles it."""
    from euispice_coreg_amd import parallel
    assert parallel.block_grid(60, 60, 8) in ((4, 2), (2, 4)) and parallel.block_grid(60, 60, 4) == (2, 2)
    assert parallel.block_grid(121, 1, 8) == (8, 1)
    for shape5 in [(60, 60, 1, 1, 1), (61, 7, 2, 1, 3), (5, 9, 1, 1, 1), (3, 1, 1, 2, 1)]:
        full = np.arange(int(np.prod(shape5)), dtype=np.float64).reshape(shape5)
        for world in (1, 2, 3, 4, 8):
            perm, chunk = parallel.block_gather_index(shape5, world)
            gathered = np.full(world * chunk, np.nan)
            seen = np.zeros(shape5[:2], dtype=int)
            for r in range(world):
                lo1, hi1, lo2, hi2 = parallel.block_bounds(shape5[0], shape5[1], world, r)
                blk = full[lo1:hi1, lo2:hi2].ravel()
                assert blk.size <= chunk
                gathered[r * chunk:r * chunk + blk.size] = blk
                seen[lo1:hi1, lo2:hi2] += 1
            assert (seen == 1).all(), (shape5, world)
            assert np.array_equal(gathered[perm], full.ravel()), (shape5, world)


def test_lag_sharding_modes():
    """Which partition a sweep gets (parallel.lag_sharding): the drop-in API and bench.py decide with this function."""
    from euispice_coreg_amd import parallel
    assert parallel.lag_sharding((60, 60, 1, 1, 1), 1) == "none"
    for world in (2, 4, 8):
        assert parallel.lag_sharding((60, 60, 1, 1, 1), world) == "blocks"       # headline
        assert parallel.lag_sharding((121, 121, 1, 1, 1), world) == "blocks"     # cfg3
        assert parallel.lag_sharding((1, 2000, 1, 1, 1), world) == "blocks"      # one axis: blocks along the other
        assert parallel.lag_sharding((5, 5, 1, 1, 1), world) == "points"         # few lag-points per GPU
        # 3-D / 5-D sweeps: the (cdelt1, cdelt2, crota) combinations are dealt to the GPUs, each sweeps the
        # whole plane at one-GPU efficiency
        assert parallel.lag_plan((41, 41, 5, 5, 11), world) == ("combos", world, 1, 1)   # cfg5
        assert parallel.lag_plan((1, 1, 1, 1, 2001), world) == ("combos", world, 1, 1)   # a pure CROTA sweep
    assert parallel.lag_plan((61, 61, 1, 1, 21), 8) == ("combos", 8, 1, 1)               # cfg4
    assert parallel.lag_plan((61, 61, 1, 1, 21), 4)[0] == "combos" and parallel.lag_sharding((61, 61, 1, 1, 21), 2) in (
        "blocks", "combos")
    assert parallel.lag_plan((3, 3, 5, 5, 11), 8) == ("combos", 8, 1, 1)
    assert parallel.lag_plan((3, 3, 1, 1, 2), 8) == ("points", 1, 1, 1)
    # 7 GPUs, 6 combinations, and a 12 x 15 plane that no 7 x 1 / 1 x 7 grid of ceil-sized blocks covers without an
    # empty one: the literal np.array_split of the raveled index (alignment.py:677-687)
    assert parallel.lag_sharding((12, 15, 1, 1, 6), 7) == "slices"
    # balanced runs of combinations
    assert [parallel.combo_bounds(21, 8, k) for k in range(8)] == [(0, 3), (3, 6), (6, 9), (9, 12), (12, 15), (15, 17),
                                                                   (17, 19), (19, 21)]
    assert parallel.lag_batches(60, 60) == 15 and parallel.lag_batches(41, 41) == 7 and parallel.lag_batches(21, 11) == 1


def test_grid_gather_index_covers_every_lag_point_once():
    """parallel.grid_gather_index: the permutation that turns the all-gathered shares (block of the plane x run of
    combinations, padded to one chunk per rank) into the C-order map, for block, combo and mixed plans."""
    from euispice_coreg_amd import parallel
    for shape5 in ((7, 5, 1, 1, 1), (24, 24, 1, 1, 3), (20, 20, 1, 1, 3), (6, 9, 2, 3, 2), (41, 41, 5, 5, 11),
                   (61, 61, 1, 1, 21), (1, 1, 1, 1, 2001)):
        full = np.arange(int(np.prod(shape5)), dtype=np.float64).reshape(shape5)
        inner = shape5[2] * shape5[3] * shape5[4]
        for world in (2, 3, 4, 8):
            mode, gc, g1, g2 = parallel.lag_plan(shape5, world)
            if mode not in ("blocks", "combos"):
                continue
            assert gc * g1 * g2 == world
            perm, chunk = parallel.grid_gather_index(shape5, world)
            gathered = np.full(chunk * world, np.nan)
            for r in range(world):
                lo1, hi1, lo2, hi2, c_lo, c_hi = parallel.grid_share(shape5, world, r)
                assert hi1 > lo1 and hi2 > lo2
                part = full.reshape(shape5[0], shape5[1], inner)[lo1:hi1, lo2:hi2, c_lo:c_hi].ravel()
                assert part.size <= chunk
                gathered[r * chunk:r * chunk + part.size] = part
            assert np.array_equal(gathered[perm], full.ravel()), (shape5, world)
            if gc == 1:
                p2, c2 = parallel.block_gather_index(shape5, world)
                assert c2 == chunk and np.array_equal(p2, perm)


def test_allgather_lag_slices_gloo_world2(tmp_path):
    """N > 1 path on CPU: two ranks (gloo), each 'sweeps' its slice, one all-gather assembles the map."""
    script = tmp_path / "worker.py"
    script.write_text(
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch.distributed as dist\n"
        "from euispice_coreg_amd import parallel\n"
        "dist.init_process_group('gloo')\n"
        "rank, world = parallel.world_info()\n"
        "for n in (1, 5, 121, 3600):\n"
        "    lo, hi, chunk = parallel.shard_bounds(n, world, rank)\n"
        "    local = np.arange(lo, hi, dtype=np.float64) * 0.5 + 1.0\n"
        "    full = parallel.allgather_lag_slices(local, n).numpy()\n"
        "    assert np.array_equal(full, np.arange(n) * 0.5 + 1.0), (rank, n, full)\n"
        "modes = set()\n"
        "for shape5 in [(60, 60, 1, 1, 1), (280, 120, 2, 1, 3), (40, 360, 1, 1, 2), (24, 24, 1, 1, 2), (24, 24, 1, 1, 3),\n"
        "               (41, 41, 5, 5, 11), (1, 1, 1, 1, 2001), (1, 300, 1, 1, 1)]:\n"
        "    mode = parallel.lag_sharding(shape5, world)\n"
        "    assert mode in ('blocks', 'combos')\n"
        "    modes.add(mode)\n"
        "    inner = shape5[2] * shape5[3] * shape5[4]\n"
        "    want = np.arange(int(np.prod(shape5)), dtype=np.float64).reshape(shape5[0], shape5[1], inner) * 0.25 - 3.0\n"
        "    lo1, hi1, lo2, hi2, c_lo, c_hi = parallel.grid_share(shape5, world, rank)\n"
        "    full = parallel.allgather_lag_blocks(want[lo1:hi1, lo2:hi2, c_lo:c_hi], shape5)\n"
        "    assert np.array_equal(full, want.ravel()), (rank, shape5)\n"
        "assert modes == {'blocks', 'combos'}\n"
        "assert parallel.replicate_image(np.zeros((4, 4), dtype=np.float32)) is None  # gloo: the caller uploads\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "print('rank', rank, 'ok')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29561", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_jitter_sublists_follow_reference_formula():
    """jitter_correction.py:91-98."""
    from euispice_coreg_amd.jitter_correction.jitter_correction import build_sublists, _time_tag
    after, before = build_sublists(25, 10, 1)
    assert [a.tolist() for a in after] == [list(range(0, 11)), list(range(10, 21)), list(range(20, 25))]
    assert [b.tolist() for b in before] == [[0]]
    after, _ = build_sublists(7, 3, 2)
    assert [a.tolist() for a in after] == [[0, 1, 2, 3, 4], [3, 4, 5, 6], [6]]
    assert _time_tag("2022-03-17T09:50:45.277") == "09_50_45"


def test_jitter_rejects_zero_overlap(tmp_path):
    from euispice_coreg_amd.jitter_correction import jitter_correction_imagers
    with pytest.raises(ValueError):
        jitter_correction_imagers([], str(tmp_path), overlap=0)


def test_fits_header_only_read_and_identity(tmp_path):
    from euispice_coreg_amd.utils import fits_io
    p = str(tmp_path / "a.fits")
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    fits_io.write_images(p, [(None, {"ORIGIN": "x"}), (img, {"CRVAL1": 1.5, "EXTNAME": "IMG", "DATE-AVG": "2022-03-17T09:50:45.277"})])
    h = fits_io.read_header(p, -1)
    assert h["CRVAL1"] == 1.5 and h["NAXIS1"] == 4 and h["DATE-AVG"].startswith("2022-03-17T09:50:45")
    assert fits_io.read_header(p, "IMG")["NAXIS2"] == 3
    d, h2 = fits_io.read_image(p, -1)
    assert d.dtype == np.float32 and np.array_equal(d, img) and h2 == h
    ident = fits_io.file_identity(p, -1)
    assert ident == fits_io.file_identity(p, -1) and ident[0] == p
    assert fits_io.file_identity((img, {}), -1) is None


def test_fits_reader_decodes_the_selected_hdu_of_several(tmp_path):
    """read_image decodes only the HDU asked for (index from either end, or EXTNAME) of a file with several image HDUs of
    different types; BSCALE / BZERO scaling (written by hand: the writer never emits it) gives float64 like astropy."""
    from euispice_coreg_amd.utils import fits_io
    p = str(tmp_path / "m.fits")
    a = (np.arange(35, dtype=np.float32).reshape(5, 7) - 3.5) * 1.25
    b = np.arange(24, dtype=np.int16).reshape(4, 6) - 7
    c = np.linspace(-1.0, 1.0, 6).reshape(2, 3)
    fits_io.write_images(p, [(None, {}), (a, {"EXTNAME": "A"}), (b, {"EXTNAME": "B"}), (c, {"EXTNAME": "C"})])
    for win, want in ((1, a), ("A", a), (-3, a), (2, b), ("B", b), (-1, c), (3, c), ("C", c)):
        d, h = fits_io.read_image(p, win)
        assert d.dtype == want.dtype and d.dtype.isnative and np.array_equal(d, want), win
        assert h["NAXIS1"] == want.shape[1] and h["NAXIS2"] == want.shape[0]
    everything = fits_io.read_all(p)
    assert everything[0][0] is None and [None if d is None else d.shape for d, _ in everything[1:]] == [a.shape, b.shape, c.shape]
    with pytest.raises((KeyError, IndexError)):
        fits_io.read_image(p, "nope")
    with pytest.raises(IndexError):
        fits_io.read_image(p, 7)
    with pytest.raises(ValueError):
        fits_io.read_image(p, 0)  # the primary HDU holds no image
    # BSCALE / BZERO: patch the int16 HDU's header in place (its cards are padded to a whole block)
    raw = bytearray(open(p, "rb").read())
    i = raw.index(b"EXTNAME = 'B")
    end = raw.index(b"END" + b" " * 77, i)
    cards = ("BSCALE  = %20s" % "0.5").ljust(80) + ("BZERO   = %20s" % "100.0").ljust(80) + "END".ljust(80)
    assert raw[end + 80:end + 80 + 160] == b" " * 160  # room left in the block
    raw[end:end + len(cards)] = cards.encode("ascii")
    p2 = str(tmp_path / "s.fits")
    open(p2, "wb").write(bytes(raw))
    d, h = fits_io.read_image(p2, "B")
    assert d.dtype == np.float64 and np.array_equal(d, b.astype(np.float64) * 0.5 + 100.0) and h["BZERO"] == 100.0
    assert np.array_equal(fits_io.read_image(p2, "C")[0], c)


def test_jitter_session_spreads_images_over_ranks_gloo_world2(tmp_path):
    """N > 1 schedule of the jitter session on CPU (gloo, 2 ranks): images of a sublist are dealt round-robin, ranks
    meet between sublists, every corrected file is written exactly once.  The GPU sweep is replaced by a stand-in
    that returns a correlation map peaked at a lag derived from the image index."""
    from euispice_coreg_amd.utils import fits_io
    paths = []
    for k in range(7):
        p = str(tmp_path / f"f{k}.fits")
        hdr = {"CRVAL1": 10.0, "CRVAL2": -5.0, "CDELT1": 1.0, "CDELT2": 1.0, "CUNIT1": "arcsec", "CUNIT2": "arcsec",
               "CROTA": 0.0, "CRPIX1": 2.0, "CRPIX2": 2.0, "DATE-AVG": "2022-03-17T09:50:%02d.000" % (10 + k)}
        fits_io.write_images(p, [(None, {}), (np.full((3, 4), float(k), dtype=np.float32), hdr)])
        paths.append(p)
    out = str(tmp_path / "out")
    script = tmp_path / "worker.py"
    script.write_text(
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch.distributed as dist\n"
        "from euispice_coreg_amd.jitter_correction import jitter_correction as J\n"
        "from euispice_coreg_amd.hdrshift import AlignmentResults\n"
        "dist.init_process_group('gloo')\n"
        "rank = dist.get_rank()\n"
        "lag = np.arange(-4.0, 4.5, 0.5)\n"
        "class FakeA:\n"
        "    def __init__(self, path): self.path = path\n"
        "    def _wrap(self, corr, rt, restore_units):\n"
        "        return AlignmentResults(corr, lag, lag, None, None, None, 'arcsec', image_to_align_path=self.path,\n"
        "                                image_to_align_window=-1)\n"
        "def fake(small_fov_path=None, large_fov_fits_path=None, _preloaded_small=None, **kw):\n"
        "    assert os.path.isfile(large_fov_fits_path), large_fov_fits_path  # the sublist's reference is on disk\n"
        "    k = float(np.asarray(_preloaded_small[0])[0, 0])\n"
        "    x, y = np.meshgrid(lag, lag, indexing='ij')\n"
        "    corr = 0.9 * np.exp(-((x - 0.5 * k) ** 2 + (y + 0.25 * k) ** 2) / 4.0)\n"
        "    return FakeA(small_fov_path), corr.reshape(len(lag), len(lag), 1, 1, 1, 1)\n"
        "J._align_hrieuv_with_hrieuv = fake\n"
        f"done = J.jitter_correction_imagers({paths!r}, {out!r}, lag_crval1=lag, lag_crval2=lag, sublist_length=3,\n"
        "                                    overlap=1)\n"
        f"open(os.path.join({str(tmp_path)!r}, 'rank%d.txt' % rank), 'w').write(str(sorted(i for i, _, _ in done)))\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29563")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29563", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # sublists [0,1,2,3] [3,4,5,6] [6]: rank 0 gets positions 0, 2 of each, rank 1 position 1
    assert (tmp_path / "rank0.txt").read_text() == "[1, 3, 4, 6]"
    assert (tmp_path / "rank1.txt").read_text() == "[2, 5]"
    for k in range(7):
        h = fits_io.read_header(os.path.join(out, f"f{k}.fits"), -1)
        want = 10.0 + (0.5 * k if k else 0.0)
        assert abs(h["CRVAL1"] - want) < 0.05, (k, h["CRVAL1"])
        assert abs(h["CRVAL2"] - (-5.0 - (0.25 * k if k else 0.0))) < 0.05


def test_plot_correlation_writes_a_figure(tmp_path):
    """AlignmentResults.plot_correlation (reference: plot/plot.py:56-175) on the reference's own correlation fixture."""
    pytest.importorskip("matplotlib")
    from euispice_coreg_amd.hdrshift import AlignmentResults
    R = AlignmentResults(corr=REF_CORR, lag_crval1=np.arange(15, 26, 1), lag_crval2=np.arange(5, 11, 1),
                         lag_cdelt1=None, lag_cdelt2=[0], lag_crota=[0.75], unit_lag="arcsec")
    out = tmp_path / "corr.png"
    fig, ax = R.plot_correlation(path_save_figure=str(out))
    assert out.stat().st_size > 1000
    assert ax.get_xlabel() == "CRVAL1 [arcsec]" and len(ax.images) == 1
    assert ax.images[0].get_array().shape == (6, 11)


def _run_bench(extra_env, *argv, timeout=600):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):  # a plain driver command: no torchrun
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout  # ONE JSON line, nothing else on stdout
    import json
    return json.loads(lines[0])


def test_bench_self_launches_its_ranks_gloo_world2():
    """`python bench.py --gpus 2` with no torchrun environment starts its two ranks itself (children, before any GPU
    call), runs the process group + the one all-gather + the block permutation, and rank 0 prints ONE JSON line.  Dry
    run: no GPU on this box, each rank fills its lag block with the raveled lag indices instead of sweeping."""
    out = _run_bench({"COREG_BENCH_BACKEND": "gloo", "COREG_BENCH_DRY": "1"}, "--gpus", "2", "--steps", "3", "--warmup",
                     "1")
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["dry_run"] is True and out["value"] is None
    assert out["dry_run_map_ok"] is True  # gathered blocks + permutation == C-order raveled lag map
    assert out["dry_run_pool_ok"] is True and "error" not in out  # a forked pool's SIGTERM is not this rank's
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and sum(r["lags"] for r in out["per_rank"]) == 3600
    assert out["scaling"] == "strong" and out["config"]["resident"] is True


def _run_bench_failing(extra_env, *argv, timeout=240):
    import json
    import time
    env = dict(os.environ, COREG_BENCH_BACKEND="gloo", COREG_BENCH_DRY="1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])  # still ONE line, nothing else on stdout
    return p.returncode, json.loads(lines[0]), time.time() - t0


@pytest.mark.parametrize("where", ["init", "step"])
def test_bench_says_so_on_its_one_line_when_a_rank_dies(where):
    """`python bench.py --gpus 2` with rank 1 made to exit (before the process group exists / after
    the warm-up, i.e. the others are inside or about to enter a collective): the launcher comes back non-zero within the
    deadline and stdout holds ONE JSON line with "error", value null and the number of ranks that joined the group.  The
    reference's counterpart is the unbounded busy-wait join of alignment.py:723-744."""
    rc, out, dt = _run_bench_failing({"COREG_BENCH_TEST_FAIL_RANK": "1", "COREG_BENCH_TEST_FAIL_AT": where,
                                      "COREG_BENCH_DEADLINE_S": "120", "COREG_BENCH_PG_TIMEOUT_S": "20"},
                                     "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert rc != 0 and dt < 120
    assert out["value"] is None and out["error"] and out["n_gpus"] == 2
    assert out["n_ranks_seen"] == (2 if where == "step" else out["n_ranks_seen"]) and 0 <= out["n_ranks_seen"] <= 2
    assert out["metric"] and out["unit"] == "lag-points/s"


def test_bench_deadline_ends_a_run_whose_rank_never_joins_the_collective():
    """A rank that stays out of a collective for ever: the process-group timeout (here 8 s) or the ranks' own deadline
    (here 40 s) ends the run; ONE error line, non-zero return code, no process left behind."""
    rc, out, dt = _run_bench_failing({"COREG_BENCH_TEST_FAIL_RANK": "1", "COREG_BENCH_TEST_FAIL_AT": "hang",
                                      "COREG_BENCH_DEADLINE_S": "40", "COREG_BENCH_PG_TIMEOUT_S": "8"},
                                     "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert rc != 0 and dt < 100
    assert out["value"] is None and out["error"] and out["n_ranks_seen"] == 2


def _several_hdus(tmp_path):
    """A file with an empty primary, a float32 image, an int16 image with BSCALE / BZERO and a float64 image."""
    from euispice_coreg_amd.utils import fits_io
    from tests import helpers as H
    small, hs, _, _, _ = H.scene(small_n=40, large_n=32)
    rng = np.random.default_rng(5)
    h16 = dict(hs, EXTNAME="RAW16", BSCALX=0.25, BZERX=100.0)  # (write_images drops the real keywords: renamed below)
    p = str(tmp_path / "several.fits")
    fits_io.write_images(p, [(None, {"ORIGIN": "test"}), (small.astype(np.float32), dict(hs, EXTNAME="IMAGE")),
                             (rng.integers(-3000, 3000, (40, 40)).astype(np.int16), h16),
                             (small.astype(np.float64) * (1 + 1e-9), dict(hs, EXTNAME="F64"))])
    blob = open(p, "rb").read()
    assert blob.count(b"BSCALX  =") == 1 and blob.count(b"BZERX   =") == 1
    open(p, "wb").write(blob.replace(b"BSCALX  =", b"BSCALE  =").replace(b"BZERX   =", b"BZERO   ="))
    return p, hs


def test_raw_image_is_the_file_s_bytes_and_decodes_like_read_image(tmp_path):
    from euispice_coreg_amd.utils import fits_io
    p, _ = _several_hdus(tmp_path)
    whole = open(p, "rb").read()
    for window in (1, 2, 3, -1, "IMAGE"):
        raw = fits_io.open_raw(p, window)
        data, hdr = fits_io._read_all(p, only=window)[fits_io._select(
            [(h, None) for h in fits_io._scan(p)[0]], window)][::-1]
        assert raw is not None and raw.shape == data.shape and raw.header == hdr
        got = np.asarray(raw)
        assert got.dtype == data.dtype == raw.dtype and np.array_equal(got, data, equal_nan=True)
        # the mapped bytes are the data unit itself
        mapped = bytes(np.frombuffer(raw._bytes, dtype=np.uint8))
        assert whole.count(mapped) >= 1 and len(mapped) == data.size * abs(raw.bitpix) // 8
        assert fits_io.native_pixels(raw) is raw
        raw.close()
    assert fits_io.open_raw(p, 2).bscale == 0.25 and fits_io.open_raw(p, 2).bzero == 100.0
    assert fits_io.open_raw(p, 0) is None            # no data unit
    assert fits_io.open_raw((np.zeros((2, 2)), {}), 0) is None
    assert fits_io.open_raw(str(tmp_path / "missing.fits")) is None
    assert fits_io.open_raw(p, 9) is None and fits_io.open_raw(p, "NOPE") is None


def test_write_corrected_fits_copies_data_units_and_rewrites_only_the_selected_headers(tmp_path):
    """utils/Util.py:106-159: selected windows get corrected keywords and float32 data, everything else is copied."""
    from euispice_coreg_amd.hdrshift import AlignmentResults
    from euispice_coreg_amd.utils import fits_io, header as hdrutil
    p, hs = _several_hdus(tmp_path)
    R = AlignmentResults(REF_CORR, np.arange(15, 26, 1), np.arange(5, 11, 1), None, [0], [0.75], "arcsec",
                         image_to_align_path=p)
    out = str(tmp_path / "out.fits")
    R.write_corrected_fits(["IMAGE", 2], out)
    src, dst = open(p, "rb").read(), open(out, "rb").read()
    (h_in, sp_in), (h_out, sp_out) = fits_io._scan(p), fits_io._scan(out)
    assert len(h_out) == 4
    # HDU 0 and 3: byte for byte, header included
    pad = lambda n: (n + 2879) // 2880 * 2880
    assert dst[:2880] == src[:2880]
    a, b = sp_in[2][0] + pad(sp_in[2][1]), sp_out[2][0] + pad(sp_out[2][1])   # where HDU 3's header starts
    assert src[a:] == dst[b:] and len(src) - a > 2880 + 40 * 40 * 8
    # HDU 1 (float32): the data unit is the input's, bit for bit
    n = sp_in[1][1]
    assert src[sp_in[1][0]:sp_in[1][0] + n] == dst[sp_out[1][0]:sp_out[1][0] + n]
    # HDU 2 (int16, scaled): converted to float32 as np.array(data, dtype="<f4") does
    want = np.array(np.asarray(fits_io.open_raw(p, 2)), dtype="<f4")
    got, h2 = fits_io.read_image(out, 2)
    assert got.dtype == np.float32 and np.array_equal(got, want) and "BSCALE" not in h2 and h2["BITPIX"] == -32
    # the corrected headers are what the decode -> correct -> encode path writes
    ref_out = str(tmp_path / "ref.fits")
    hdus = []
    for ii, (d, h) in enumerate(fits_io.read_all(p)):
        if ii in (1, 2):
            h = h.copy()
            s = R.shift_arcsec
            hdrutil.correct_pointing_header(h, lag_crval1=s[0], lag_crval2=s[1], lag_cdelt1=s[2], lag_cdelt2=s[3],
                                            lag_crota=s[4])
            d = np.array(d, dtype="<f4")
        hdus.append((d, h))
    fits_io.write_images(ref_out, hdus)
    h_ref = fits_io._scan(ref_out)[0]
    assert h_out[1] == h_ref[1] and h_out[2] == h_ref[2]
    assert h_out[1]["CRVAL1"] == pytest.approx(hs["CRVAL1"] + R.shift_arcsec[0], rel=1e-14)
    # in place
    R.write_corrected_fits([1], out, path_to_l2_input=out)
    assert fits_io._scan(out)[0][1]["CRVAL1"] == pytest.approx(hs["CRVAL1"] + 2 * R.shift_arcsec[0], rel=1e-13)
    with pytest.raises(ValueError):
        R.write_corrected_fits(["no-such-window"], out)


def test_long_string_keywords_follow_the_continue_convention(tmp_path):
    """Instrument headers carry strings longer than a card (lists of parent files, processing history): the FITS
    long-string convention -- a value ending in '&' goes on in the CONTINUE cards that follow.  Read whole, written back as
    CONTINUE cards, and carried verbatim through write_corrected_fits (which copies every card it does not correct)."""
    from euispice_coreg_amd.hdrshift import AlignmentResults
    from euispice_coreg_amd.utils import fits_io
    from tests.test_oracle_golden import REF_CORR
    rng = np.random.default_rng(0)
    p = str(tmp_path / "l.fits")
    for t in range(60):
        v = "".join(rng.choice(list("ab '&/=,"), size=int(rng.integers(60, 400)))).rstrip()
        v = v + "x" if v.endswith("&") or not v else v
        hdr = {"CRVAL1": 1.0, "LONGSTR": v, "AMPERSND": "ends with &", "AFTER": 7}
        fits_io.write_images(p, [(None, {}), (np.zeros((2, 2), np.float32), hdr)])
        h = fits_io.read_header(p, -1)
        assert h["LONGSTR"] == v and h["AFTER"] == 7 and h["AMPERSND"] == "ends with &"
        raw = open(p, "rb").read()
        assert len(raw) % 2880 == 0 and (len(v.replace("'", "''")) <= 68 or b"CONTINUE  '" in raw)
    parents = ", ".join(f"solo_L1_eui-hrieuv174-image_20220317T0950{k:02d}277_V01.fits" for k in range(6))
    hdr = {"CRVAL1": 10.0, "CRVAL2": 20.0, "CDELT1": 0.5, "CDELT2": 0.5, "CROTA": 0.0, "CUNIT1": "arcsec", "CUNIT2": "arcsec",
           "CRPIX1": 3.0, "CRPIX2": 3.0, "PARENT": parents, "PC1_1": 1.0, "PC1_2": 0.0, "PC2_1": 0.0, "PC2_2": 1.0}
    src, dst = str(tmp_path / "in.fits"), str(tmp_path / "out.fits")
    fits_io.write_images(src, [(None, {}), (np.ones((6, 6), np.float32), hdr)])
    R = AlignmentResults(REF_CORR, np.arange(15, 26, 1), np.arange(5, 11, 1), None, [0], [0], "arcsec", image_to_align_path=src)
    R.write_corrected_fits([-1], dst)
    h = fits_io.read_header(dst, -1)
    assert h["PARENT"] == parents and h["CRVAL1"] == pytest.approx(10.0 + R.shift_arcsec[0])
    assert open(dst, "rb").read().count(b"CONTINUE  '") == open(src, "rb").read().count(b"CONTINUE  '") > 0


def test_write_corrected_fits_leaves_valid_checksums(tmp_path):
    """CHECKSUM / DATASUM (FITS standard 4.0, appendix J).  The fixture was written by astropy 4.3.1 with checksum=True
    (tests/golden/make_golden_checksum.py, which also had astropy verify the corrected file this package writes from it):
    every plain HDU of it adds up to -0 under the package's accumulator and its DATASUM card is the sum of its data unit
    -- that pins `_sum32` / `_encode_checksum` -- and the corrected file keeps that property: the header changed, the
    float32 data unit is the input's (DATASUM kept), the unsigned-16 one became float32 (DATASUM recomputed).  The
    reference, like astropy's default, leaves the stale cards of the input behind."""
    from euispice_coreg_amd.hdrshift import AlignmentResults
    from euispice_coreg_amd.utils import fits_io
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checksum", "three_hdus_checksum.fits")

    def hdus(path):
        raw = open(path, "rb").read()
        out = []
        with open(path, "rb") as f:
            while True:
                start = f.tell()
                h, _ = fits_io._read_header(f)
                if h is None:
                    break
                n, _shape = fits_io._data_size(h)
                pos, pad = f.tell(), (n + 2879) // 2880 * 2880
                f.seek(pad, 1)
                out.append((h, fits_io._sum32(raw[start:pos + pad]), fits_io._sum32(raw[pos:pos + pad]), raw[start:pos]))
        return out

    before = hdus(src)
    assert [h.get("EXTNAME") for h, *_ in before] == [None, "F32", "U16", "RICE"]
    for h, total, dsum, _ in before[:3]:
        assert total == 0xFFFFFFFF and int(h["DATASUM"]) == dsum
    corr = np.zeros((5, 5, 1, 1, 1, 1))
    corr[2, 3] = 1.0
    lag = np.arange(-2.0, 3.0)
    R = AlignmentResults(corr, lag, lag, None, [0], [0.5], "arcsec", image_to_align_path=src)
    dst = str(tmp_path / "corrected.fits")
    R.write_corrected_fits([1, 2, 3], dst)
    after = hdus(dst)
    for k in (0, 1, 2):
        h, total, dsum, raw_hdr = after[k]
        assert total == 0xFFFFFFFF and int(h["DATASUM"]) == dsum, h.get("EXTNAME")
    assert after[1][2] == before[1][2] and after[2][2] != before[2][2] and after[2][0]["BITPIX"] == -32
    assert after[1][0]["CHECKSUM"] != before[1][0]["CHECKSUM"] and after[1][0]["CRVAL1"] != before[1][0]["CRVAL1"]
    # comments, HISTORY and the checksum card's own comment survive the correction
    assert b"/ [arcsec] reference value" in after[1][3] and b"HISTORY made for the checksum test" in after[1][3]
    assert b"/ HDU checksum updated" in after[1][3]
    # the tile-compressed HDU: astropy 4.3.1 wrote cards that do not describe the bytes on disk; they are kept as
    # consistent as they were (header + DATASUM card add up the same way), the compressed stream is the input's
    assert after[3][2] == before[3][2]
    hb, ha = before[3][0], after[3][0]
    assert ha["DATASUM"] == hb["DATASUM"] and ha["CRVAL1"] != hb["CRVAL1"]
    assert fits_io._sum32(after[3][3], start=int(ha["DATASUM"])) == fits_io._sum32(before[3][3], start=int(hb["DATASUM"]))
    d, _ = fits_io.read_image(dst, "RICE")
    assert np.array_equal(d, fits_io.read_image(src, "RICE")[0])


def test_header_scans_are_cached_per_file_state(tmp_path):
    """A drop-in call asks for a file's headers several times: parsed once per (path, modification time, size); a file
    that is rewritten is parsed again, and callers cannot alter the cached headers."""
    from euispice_coreg_amd.utils import fits_io
    p = str(tmp_path / "a.fits")
    fits_io.write_images(p, [(None, {}), (np.zeros((3, 4), np.float32), {"CRVAL1": 1.0, "EXTNAME": "A"})])
    h1 = fits_io.read_header(p, -1)
    h1["CRVAL1"] = 99.0
    assert fits_io.read_header(p, "A")["CRVAL1"] == 1.0 and fits_io.open_raw(p, -1).header["CRVAL1"] == 1.0
    n = len(fits_io._SCAN_CACHE)
    fits_io.read_header(p, -1)
    assert len(fits_io._SCAN_CACHE) == n
    os.utime(p, ns=(1, 1))  # (same size, other modification time)
    fits_io.write_images(p, [(None, {}), (np.zeros((3, 4), np.float32), {"CRVAL1": 2.0, "EXTNAME": "A"})])
    assert fits_io.read_header(p, -1)["CRVAL1"] == 2.0
    for k in range(40):
        q = str(tmp_path / f"f{k}.fits")
        fits_io.write_images(q, [(None, {"K": k})])
        assert fits_io.read_header(q, 0)["K"] == k
    assert len(fits_io._SCAN_CACHE) <= 16


def test_gzip_compressed_fits_files_are_opened_transparently(tmp_path):
    """`*.fits.gz` (astropy, hence the reference, opens them as any file; `writeto` compresses by the name): headers,
    pixels, the raw and compressed-tile views the uploads use, and write_corrected_fits in and out."""
    import gzip
    from euispice_coreg_amd.hdrshift import AlignmentResults
    from euispice_coreg_amd.utils import fits_io
    from tests.test_oracle_golden import REF_CORR
    hdr = {"CRVAL1": 10.0, "CRVAL2": 20.0, "CDELT1": 0.5, "CDELT2": 0.5, "CROTA": 0.0, "CUNIT1": "arcsec", "CUNIT2": "arcsec",
           "CRPIX1": 3.0, "CRPIX2": 3.0, "PC1_1": 1.0, "PC1_2": 0.0, "PC2_1": 0.0, "PC2_2": 1.0, "EXTNAME": "IMG"}
    img = np.arange(30, dtype=np.float32).reshape(5, 6)
    plain, gz = str(tmp_path / "a.fits"), str(tmp_path / "a.fits.gz")
    fits_io.write_images(plain, [(None, {}), (img, hdr)])
    with open(plain, "rb") as fi, gzip.open(gz, "wb") as fo:
        fo.write(fi.read())
    assert fits_io.read_header(gz, "IMG")["CRVAL1"] == 10.0
    assert np.array_equal(fits_io.read_image(gz, -1)[0], img)
    raw = fits_io.open_raw(gz, -1)
    assert raw is not None and np.array_equal(np.asarray(raw), img)
    up, h = fits_io.load_for_upload(gz, -1)
    assert isinstance(up, fits_io.RawImage) and h["CRVAL2"] == 20.0
    # a tile-compressed image inside a gzip file
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "compressed", "rice_i16.fits")
    gz2 = str(tmp_path / "rice.fits.gz")
    with open(src, "rb") as fi, gzip.open(gz2, "wb") as fo:
        fo.write(fi.read())
    ci = fits_io.open_compressed(gz2, -1)
    assert ci is not None and ci.on_gpu and np.array_equal(np.asarray(ci), np.asarray(fits_io.open_compressed(src, -1)))
    # corrected files: gzip in, plain out and gzip out
    R = AlignmentResults(REF_CORR, np.arange(15, 26, 1), np.arange(5, 11, 1), None, [0], [0], "arcsec", image_to_align_path=gz)
    out_plain, out_gz = str(tmp_path / "c.fits"), str(tmp_path / "c.fits.gz")
    R.write_corrected_fits([-1], out_plain)
    R.write_corrected_fits([-1], out_gz)
    assert open(out_gz, "rb").read(2) == bytes([0x1F, 0x8B]) and open(out_plain, "rb").read(6) == b"SIMPLE"
    assert gzip.open(out_gz, "rb").read() == open(out_plain, "rb").read()
    assert fits_io.read_header(out_gz, -1)["CRVAL1"] == pytest.approx(10.0 + R.shift_arcsec[0])
    assert np.array_equal(fits_io.read_image(out_gz, -1)[0], img)


def test_bench_counted_flop_of_every_variant_and_config_arguments():
    """`bench.py --config`: the counted float64 flop per (point, lag) of each sweep variant follows the headline's
    convention (42 for TRANSLATE order 2), and a variant other than the headline refuses N > 1."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("coreg_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b._variant_flop("TRANSLATE", 2, False) == b.FLOP_PER_POINT_LAG == 42.0
    assert b._variant_flop("TRANSLATE", 1, False) == 23.0 and b._variant_flop("TRANSLATE", 3, False) == 78.0
    assert b._variant_flop("HOMOGRAPHY_SERIES", 2, True) == 60.0 and b._variant_flop("HOMOGRAPHY", 2, True) == 62.0
    assert b._variant_flop("CAR", 2, True) == 72.0
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg2", "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "one GPU" in (p.stderr + p.stdout)
    line = json_line = b.error_line("x")
    import json
    d = json.loads(line)
    assert d["value"] is None and d["error"] == "x" and d["unit"] == "lag-points/s" and json_line
