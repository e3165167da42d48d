"""GPU test of the synthetic-raster builder (euispice_coreg_amd.synras.map_builder, reference:
synras/map_builder.py:89-214, 251-349) against a column-by-column restatement with the oracle's TAN WCS and
scipy's map_coordinates."""
import datetime as dt
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def raster(tmp_path_factory):
    from euispice_coreg_amd import synthetic
    from euispice_coreg_amd.utils import fits_io
    d = tmp_path_factory.mktemp("synras")
    cube, h4, large, hl, truth = synthetic.make_spice_l2()
    frames = synthetic.make_imager_sequence(large, hl)
    p_spice = str(d / "solo_L2_spice-n-ras_20220317T094045_V01.fits")
    fits_io.write_images(p_spice, [(cube, h4)])
    paths = []
    for k, (img, h) in enumerate(frames):
        p = str(d / f"solo_L2_eui-fsi174-image_{k:02d}.fits")
        fits_io.write_images(p, [(None, {}), (img, h)])
        paths.append(p)
    return p_spice, paths, frames, h4, d


def _reference_run():
    """What the REFERENCE's own SPICEComposedMapBuilder.process did with the same raster geometry and imager sequence
    (tests/golden/make_golden_callers.py -> callers_golden.json): the imager frame it took for every raster column and
    the 2-D header it composed.  (the expectation below used to be built from this package's own
    `spice_header.column_times` / `celestial_header`.)"""
    from tests.test_reference_callers_cpu import load
    _, m = load()
    c = m["synras"]["cases"]["process"]
    return c["frame_of_column"], c["header"]


def _expected(h4, frames, target):
    """Literal per-column construction (map_builder.py:95-131) on the CPU: the oracle's TAN WCS + scipy, the column ->
    frame choice and the target header taken from the reference's own run."""
    from oracle import coreg_oracle as O
    frame_of_column, _ = _reference_run()
    ny, nx = target["NAXIS2"], target["NAXIS1"]
    assert len(frame_of_column) == nx
    wt = O.TanWCS(target)
    out = np.empty((ny, nx))
    chosen = []
    for ii in range(nx):
        k = int(frame_of_column[ii])
        chosen.append(k)
        img, hi = frames[k]
        hi = dict(hi)
        O.check_and_create_pcij_matrix(hi)
        lon, lat = wt.pixel_to_world(np.full(ny, float(ii)), np.arange(ny, dtype=np.float64))
        px, py = O.TanWCS(hi).world_to_pixel(O.ang2pipi(lon), O.ang2pipi(lat))
        dst = np.empty(ny, dtype=np.float32)
        out[:, ii] = O.interpol2d(img, px, py, fill=np.nan, order=2, dst=dst)
    return out, chosen


def test_spice_composed_map_matches_column_by_column_construction(raster, tmp_path):
    from euispice_coreg_amd.synras.map_builder import SPICEComposedMapBuilder
    from euispice_coreg_amd.utils import fits_io, spice_header as S
    p_spice, paths, frames, h4, _ = raster
    C = SPICEComposedMapBuilder(path_to_spectro=p_spice, list_imager_paths=paths, threshold_time=200.0,
                                window_imager=-1, window_spectro=0)
    name = C.process(folder_path_output=str(tmp_path), basename_output="synras.fits", print_filename=False,
                     return_synras_name=True)
    assert name == os.path.join(str(tmp_path), "synras.fits") == C.get_path_to_composed_map()
    data, hdr = fits_io.read_image(name, 0)
    # the target WCS: the cards the reference composed for the same raster (CRVAL / CDELT / PC / CRPIX in degrees)
    ref_hdr = _reference_run()[1]
    target = {k: ref_hdr[k] for k in ("CRPIX1", "CRPIX2", "CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "PC1_1", "PC1_2", "PC2_1",
                                      "PC2_2", "CUNIT1", "CUNIT2", "CTYPE1", "CTYPE2")}
    target.update(NAXIS1=h4["NAXIS1"], NAXIS2=h4["NAXIS2"], LONPOLE=ref_hdr.get("LONPOLE", 180.0))
    flat = dict(S.celestial_header(h4))  # (and this package's flattening agrees with it, card for card)
    assert all(flat[k] == v for k, v in target.items() if not k.startswith("NAXIS")), target
    want, chosen = _expected(h4, frames, target)
    assert len(set(chosen)) >= 4  # the raster spans several imager frames
    assert data.shape == want.shape
    assert np.array_equal(np.isnan(data), np.isnan(want))
    # float32-rounded samples of a float64 accumulation: equal up to one float32 ulp of rounding ties
    assert np.nanmax(np.abs(data - want) / np.abs(want)) <= 1.3e-7
    assert (data == want)[np.isfinite(want)].mean() > 0.999
    # header: imager frame of the middle step + SPICE pointing keywords (in degrees after flattening)
    assert hdr["CUNIT1"] == "deg" and abs(hdr["CRVAL1"] * 3600 - h4["CRVAL1"]) < 1e-8
    assert hdr["CDELT2"] == target["CDELT2"] and hdr["PC1_2"] == target["PC1_2"]
    assert hdr["DATE-AVG"] == h4["DATE-AVG"] and hdr["SPECPATH"] == os.path.basename(p_spice)
    assert hdr["DETECTOR"] == "FSI" and hdr["WAVELNTH"] == 174
    # usable as the reference of AlignmentSpice: aligning the raster on its own synthetic raster finds the pointing
    from euispice_coreg_amd.hdrshift import AlignmentSpice
    lag = np.arange(-30.0, -14.0, 2.5)
    A = AlignmentSpice(name, p_spice, lag_crval1=lag, lag_crval2=np.arange(28.0, 45.0, 2.5), large_fov_window=0,
                       small_fov_window=0, parallelism=True)
    with pytest.warns(UserWarning):
        res = A.align_using_helioprojective()
    assert np.isfinite(res.corr).any()


def test_threshold_time_and_in_memory_header(raster):
    from euispice_coreg_amd.synras.map_builder import SPICEComposedMapBuilder
    p_spice, paths, frames, h4, _ = raster
    with pytest.raises(ValueError):
        SPICEComposedMapBuilder(p_spice, paths[:2], threshold_time=100.0, window_spectro=0).process(
            folder_path_output=None, print_filename=False)
    C = SPICEComposedMapBuilder(p_spice, paths, threshold_time=200.0, window_spectro=0)
    C.process_from_header(h4)
    assert C.hdr_composed["NAXIS1"] == h4["NAXIS1"] and C.data_composed.shape == (h4["NAXIS2"], h4["NAXIS1"])
    # keep_original_imager_pixel_size: sampled every CDELT_imager, header re-centred (map_builder.py:163-189)
    C.process_from_header(h4, keep_original_imager_pixel_size=True)
    r1, r2 = 4.44 / 4.0, 4.44 / 1.098
    assert C.data_composed.shape == (len(np.arange(0, h4["NAXIS2"], r2)), len(np.arange(0, h4["NAXIS1"], r1)))
    assert abs(C.hdr_composed["CDELT1"] * 3600 - 4.44) < 1e-9 and abs(C.hdr_composed["CDELT2"] * 3600 - 4.44) < 1e-9
    assert C.hdr_composed["CRPIX1"] == (C.data_composed.shape[1] + 1) / 2
    assert np.isfinite(C.data_composed).mean() > 0.9


def _as_level3(h4, ncoef=4):
    """The same window as a level-3 coefficient map: axes (coefficient, x, y, time), map_builder.py:290-346."""
    h = {k: v for k, v in h4.items() if not any(k.startswith(p) for p in ("CRPIX", "CRVAL", "CDELT", "CUNIT", "CTYPE",
                                                                          "PC", "NAXIS"))}
    h.update(NAXIS=4, NAXIS1=ncoef, NAXIS2=h4["NAXIS1"], NAXIS3=h4["NAXIS2"], NAXIS4=1, CTYPE1="COEFF", CUNIT1="",
             CRPIX1=1.0, CRVAL1=1.0, CDELT1=1.0, CTYPE4="UTC")
    for new, old in ((2, 1), (3, 2)):
        for k in ("CRPIX", "CRVAL", "CDELT", "CUNIT", "CTYPE"):
            h[f"{k}{new}"] = h4[f"{k}{old}"]
    for k in ("CRPIX", "CRVAL", "CDELT", "CUNIT"):
        h[f"{k}4"] = h4[f"{k}4"]
    m = {1: 2, 2: 3, 4: 4}
    for i in (1, 2, 4):
        for j in (1, 2, 4):
            if f"PC{i}_{j}" in h4:
                h[f"PC{m[i]}_{m[j]}"] = h4[f"PC{i}_{j}"]
    h["PC1_1"] = 1.0
    return h


def test_level3_input_gives_the_level2_raster(raster, tmp_path):
    """A level-3 coefficient map of the same window (axes coefficient, x, y, time; map_builder.py:290-346) has the same
    (HPLN, HPLT, time) geometry as its level-2 parent: same synthetic raster, same composed header."""
    from euispice_coreg_amd.synras.map_builder import SPICEComposedMapBuilder
    from euispice_coreg_amd.utils import fits_io
    p_spice, paths, frames, h4, d = raster
    h3 = _as_level3(h4)
    p3 = str(d / "solo_L3_spice-n-ras_20220317T094045_V01.fits")
    fits_io.write_images(p3, [(np.zeros((1, h4["NAXIS2"], h4["NAXIS1"], 4), dtype=np.float32), h3)])
    C2 = SPICEComposedMapBuilder(p_spice, paths, threshold_time=200.0, window_spectro=0)
    n2 = C2.process(str(tmp_path), "l2.fits", print_filename=False, return_synras_name=True)
    C3 = SPICEComposedMapBuilder(p3, paths, threshold_time=200.0, window_spectro=0)
    n3 = C3.process(str(tmp_path), "l3.fits", print_filename=False, level=3, return_synras_name=True)
    d2, hd2 = fits_io.read_image(n2, 0)
    d3, hd3 = fits_io.read_image(n3, 0)
    assert d3.shape == (h4["NAXIS2"], h4["NAXIS1"]) and np.array_equal(d2, d3, equal_nan=True)
    for k in ("CRVAL1", "CRVAL2", "CDELT1", "CDELT2", "CRPIX1", "CRPIX2", "PC1_2", "PC2_1"):
        assert hd2[k] == hd3[k], k
    # keep_original_imager_pixel_size uses CDELT2 / CDELT3 of the level-3 header (map_builder.py:311-313)
    C3.process(str(tmp_path), "l3k.fits", print_filename=False, level=3, keep_original_imager_pixel_size=True)
    C2.process(str(tmp_path), "l2k.fits", print_filename=False, keep_original_imager_pixel_size=True)
    a, ha = fits_io.read_image(str(tmp_path / "l3k.fits"), 0)
    b, hb = fits_io.read_image(str(tmp_path / "l2k.fits"), 0)
    assert np.array_equal(a, b, equal_nan=True) and ha["CRVAL1"] == hb["CRVAL1"] and ha["CRPIX2"] == hb["CRPIX2"]
    # a level-2 header passed as level 3 is refused, and the in-memory form is level-2 only, as in the reference
    with pytest.raises(ValueError):
        C2.process(str(tmp_path), "x.fits", print_filename=False, level=3)
    with pytest.raises(NotImplementedError):
        C3.process_from_header(h3, level=3)
