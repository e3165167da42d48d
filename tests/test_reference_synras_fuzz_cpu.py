"""CPU test of the synthetic-raster builder's slit-time logic against the RANDOM family the reference's own
`SPICEComposedMapBuilder.process` produced (tests/golden/make_golden_synras_fuzz.py ->
tests/golden/synras_fuzz_golden.{npz,json}): four random SPICE windows x imager sequences with random start, cadence
and length.  map_builder.py:95-99: mean time of each raster column (through the (x, y, t) WCS) -> nearest imager frame.
The rasters themselves: tests/test_gpu_reference_synras_fuzz.py."""
import datetime as dt
import json
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN


def load():
    g = np.load(os.path.join(GOLDEN, "synras_fuzz_golden.npz"))
    with open(os.path.join(GOLDEN, "synras_fuzz_golden.json")) as f:
        m = json.load(f)
    with open(os.path.join(GOLDEN, "spice_fuzz_golden.json")) as f:
        sp = json.load(f)["scenes"]
    return g, m, sp


def windows():
    with open(os.path.join(GOLDEN, "synras_fuzz_golden.json")) as f:
        return sorted(json.load(f)["cases"])


def test_fixture_is_what_the_generator_describes():
    g, m, sp = load()
    assert m["interpreter"]["astropy"] == "4.3.1" and len(m["cases"]) == 4
    assert {c["cadence_s"] for c in m["cases"].values()} >= {120.0, 450.0}
    assert any(c["kwargs"] for c in m["cases"].values()) and any(not c["kwargs"] for c in m["cases"].values())
    assert max(len(set(c["frame_of_column"])) for c in m["cases"].values()) >= 12
    for name, c in m["cases"].items():
        assert list(g[f"{name}/raster"].shape) == c["shape"] and len(c["imager_headers"]) == c["n_frames"]


@pytest.mark.parametrize("name", windows())
def test_raster_columns_take_the_frames_the_reference_took(name):
    from euispice_coreg_amd.utils import spice_header as S
    g, m, sp = load()
    c = m["cases"][name]
    col_s, t_ref = S.column_times(sp[name]["hdr4d"])
    dates = [S.parse_date(h["DATE-AVG"]) for h in c["imager_headers"]]
    chosen = [int(np.argmin([abs((t_ref + dt.timedelta(seconds=float(s)) - d).total_seconds()) for d in dates]))
              for s in col_s]
    if not c["kwargs"]:
        assert chosen == c["frame_of_column"]
    else:
        # keep_original_imager_pixel_size resamples the raster's x axis (map_builder.py:163-189): the frames taken are
        # those of the ORIGINAL columns, in the same order, each for a run of new columns
        ref = c["frame_of_column"]
        assert sorted(set(ref)) == sorted(set(chosen)) and ref == sorted(ref, reverse=ref[0] > ref[-1])
