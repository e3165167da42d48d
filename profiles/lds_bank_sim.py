#!/usr/bin/env python3
"""
LDS bank-conflict simulator for the gather of k_sweep (kernels.hpp gather_o2), fed with the REAL lane addresses of the
headline sweep (2048^2 Carrington grid, 60 x 60 CRVAL lags of 1 arcsec on the HRIEUV-like image: a ~2.03 px lag lattice
rolled by 3 deg).  CPU only (numpy + the oracle's Carrington coordinates).

Model (MI355X_MICROARCH.md, LDS table): a wave64 `ds_read_b64` is served in two groups of 32 lanes; a lane touches the
bank PAIR (address / 8) mod 32; lanes of a group that name the same address share one access (broadcast); every further
DISTINCT address on a busy bank pair adds one LDS cycle.  Cycles per wave-instruction = sum over the two groups of the
largest number of distinct addresses on one bank pair (conflict-free: 2).  A sample = 3 x 3 taps = 9 reads.

What is compared:
  * the lane <-> lag mapping of the plan in use (row-major 12 x 20 patch, window pitch 121) and the other pitches,
  * other patch shapes (16 x 16, 32 x 8, 8 x 32, 20 x 12) and lane orders (column-major, 8 x 4 / 4 x 8 sub-blocks),
  * lanes 0-15 of a 32-lane group take the even CRVAL1 indices of a stride-2 patch of twice the
    extent, lanes 16-31 the odd ones,
  * a SECOND copy of the window one element further (odd bank pairs), read by the lanes of odd lag rows / columns /
    (row + column) parity,
  * a de-interleaved window (even columns first, then odd columns),
  * (SIM_ONLY=pair) a row-pair layout, element (r, c) at (r >> 1) * Q + (r & 1) * R + c with Q odd, which puts rows r and
    r + 2 an odd number of elements apart (a linear pitch cannot): best 3.61 cycles (16 x 16 patch, Q = 257) -- the roll
    still mixes row parities inside a lag row -- for 4 more address instructions per sample; not built.
usage: python profiles/lds_bank_sim.py [n_points]   ->  table on stdout (committed as profiles/r03_lds_bank_sim.txt)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euispice_coreg_amd import synthetic  # noqa: E402
from oracle import coreg_oracle as O  # noqa: E402

N_PTS = int(sys.argv[1]) if len(sys.argv) > 1 else 1500


def lag_origins(hs, lag1, lag2):
    """(X0, Y0)[i1, i2] of utils/rectify.py:399-404 for every CRVAL lag (what the library puts in the lanes)."""
    roll = np.deg2rad(hs["CROTA"])
    rc, rs = np.cos(roll), np.sin(roll)
    v1 = hs["CRVAL1"] + lag1[:, None]
    v2 = hs["CRVAL2"] + lag2[None, :]
    dx = rc * v1 + rs * v2
    dy = -rs * v1 + rc * v2
    return (hs["CRPIX1"] - 1) - dx / hs["CDELT1"], (hs["CRPIX2"] - 1) - dy / hs["CDELT2"]


def wave_cycles(addr):
    """addr: int array [..., 64] of element addresses of one wave-instruction.  LDS cycles (2 = conflict-free)."""
    tot = 0
    flat = addr.reshape(-1, 64)
    out = np.zeros(flat.shape[0])
    for g in (slice(0, 32), slice(32, 64)):
        a = flat[:, g]
        worst = np.ones(flat.shape[0], dtype=int)
        bank = a & 31
        for b in range(32):
            m = bank == b
            # number of distinct addresses among the lanes on bank pair b
            aa = np.where(m, a, -1)
            aa.sort(axis=1)
            distinct = (np.diff(aa, axis=1) != 0).sum(axis=1) + 1 - (~m).any(axis=1)  # (-1 counts once when present)
            worst = np.maximum(worst, distinct)
        out += worst
    return out


def main():
    small, hs, large, hl, truth = synthetic.make_scene()
    hs = dict(hs)
    O.check_and_create_pcij_matrix(hs)
    lag = np.arange(-30, 30, 1.0)
    X0, Y0 = lag_origins(hs, lag, lag)
    # lag-independent part T of the grid points' coordinates: pixel = X0(lag) + T (rectify.py:362)
    nx, ny = O.carrington_coords(hs, 1.004, [2048, 2048], [200.0, 300.0], [-20.0, 20.0])
    x00, y00 = lag_origins(hs, np.array([0.0]), np.array([0.0]))
    T0, T1 = nx - x00[0, 0], ny - y00[0, 0]
    ok = np.isfinite(T0) & (nx > 100) & (nx < 1948) & (ny > 100) & (ny < 1948)  # interior points (the common visit)
    rng = np.random.default_rng(1)
    pick = rng.choice(np.flatnonzero(ok.ravel()), size=N_PTS, replace=False)
    t0, t1 = T0.ravel()[pick], T1.ravel()[pick]
    print(f"# {N_PTS} interior grid points, 60 x 60 lags; lag lattice: d/d(crval1) = "
          f"({X0[1,0]-X0[0,0]:+.3f}, {Y0[1,0]-Y0[0,0]:+.3f}) px, d/d(crval2) = ({X0[0,1]-X0[0,0]:+.3f}, {Y0[0,1]-Y0[0,0]:+.3f}) px")

    def lanes_of_patch(sw, sh, order="row"):
        """lag index offsets (li, lj) of the 256 lanes of one batch (padding lanes repeat lane 0)."""
        t = np.arange(256)
        if order == "row":
            li, lj = t % sw, t // sw
        elif order == "col":
            li, lj = t // sh, t % sh
        elif order in ("8x4", "4x8"):
            bw, bh = (8, 4) if order == "8x4" else (4, 8)
            blk, within = t // 32, t % 32
            nbx = max(1, sw // bw)
            li = (blk % nbx) * bw + within % bw
            lj = (blk // nbx) * bh + within // bw
        elif order == "verdict":
            # 32-lane group = one lag row of a stride-2 patch of twice the extent: lanes 0-15 even CRVAL1 indices,
            # lanes 16-31 odd ones; sw must be 32
            g, within = t // 32, t % 32
            li = np.where(within < 16, 2 * within, 2 * (within - 16) + 1)
            lj = g
        valid = (li < sw) & (lj < sh)
        return np.where(valid, li, 0), np.where(valid, lj, 0)

    def run(name, sw, sh, pitch, order="row", copy_rule=None, deinterleave=False, pair=None):
        li, lj = lanes_of_patch(sw, sh, order)
        cyc, n = 0.0, 0
        for p1 in range(0, 60, sw):
            for p2 in range(0, 60, sh):
                i1 = np.minimum(p1 + li, 59)
                i2 = np.minimum(p2 + lj, 59)
                ux = X0[i1, i2][None, :] + t0[:, None] + 0.5 - 1.0  # window-relative up to a constant origin
                uy = Y0[i1, i2][None, :] + t1[:, None] + 0.5 - 1.0
                c0 = np.floor(ux).astype(np.int64)
                r0 = np.floor(uy).astype(np.int64)
                c0 -= c0.min(axis=1, keepdims=True)
                r0 -= r0.min(axis=1, keepdims=True)
                for dr in range(3):
                    for dc in range(3):
                        c, r = c0 + dc, r0 + dr
                        if pair is not None:  # row pairs: element (r, c) at (r >> 1) * Q + (r & 1) * R + c
                            a = (r >> 1) * pair[0] + (r & 1) * pair[1] + c
                        elif deinterleave:  # even columns first, odd columns after them (half pitch each)
                            a = r * pitch + (c & 1) * (pitch // 2 + 1) + (c >> 1)
                        else:
                            a = r * pitch + c
                        if copy_rule is not None:  # second copy one element (= one bank pair) further
                            par = {"row": lj & 1, "col": li & 1, "sum": (li + lj) & 1}[copy_rule]
                            a = a + par[None, :] * (1 + 2 * 32 * 4096)  # far away, odd offset
                        for w in range(4):
                            cyc += wave_cycles(a[:, 64 * w:64 * w + 64]).sum()
                            n += a.shape[0]
        print(f"{name:58s} {cyc / n:6.2f} LDS cycles per wave read   ({(cyc / n - 2) / (cyc / n):5.1%} conflict cycles)")
        return cyc / n

    only = os.environ.get("SIM_ONLY")
    if only == "pair":
        # rows r and r + 2 an ODD number of elements apart (a linear pitch makes that distance 2 * pitch, always even,
        # which is what keeps neighbouring lag rows on the same half of the bank pairs)
        for sw, sh in ((12, 20), (16, 16), (32, 8)):
            for q, r_ in ((243, 121), (245, 122), (247, 123), (249, 124), (251, 125), (253, 126), (255, 127), (257, 128),
                          (259, 129), (261, 130), (241, 120)):
                run(f"{sw} x {sh} row-major, row pairs Q = {q}, R = {r_}", sw, sh, 0, pair=(q, r_))
        return
    if only == "copy":
        for sw, sh in ((16, 16), (12, 20), (32, 8)):
            for pitch in (128, 160):
                run(f"{sw} x {sh} row-major, pitch {pitch}, no copy", sw, sh, pitch)
                run(f"{sw} x {sh} row-major, pitch {pitch}, second copy for odd lag row", sw, sh, pitch, copy_rule="row")
        run("16 x 16 row-major, pitch 144 (16 mod 32), second copy for odd lag row", 16, 16, 144, copy_rule="row")
        return
    print("# --- the plan in use and the window pitch")
    for pitch in (113, 117, 119, 121, 123, 125):
        run(f"row-major 12 x 20 patch, pitch {pitch}", 12, 20, pitch)
    print("# --- patch shapes and lane orders (pitch 121)")
    for sw, sh in ((16, 16), (20, 12), (32, 8), (8, 32)):
        run(f"row-major {sw} x {sh} patch", sw, sh, 121)
    run("column-major 12 x 20", 12, 20, 121, "col")
    run("8 x 4 sub-blocks, 16 x 16 patch", 16, 16, 121, "8x4")
    run("4 x 8 sub-blocks, 16 x 16 patch", 16, 16, 121, "4x8")
    print("# --- VERDICT r02 item 6: even CRVAL1 indices in lanes 0-15, odd in 16-31 (32 x 8 patch)")
    for pitch in (121, 123, 125, 127, 129):
        run(f"parity-alternating 32 x 8 patch, pitch {pitch}", 32, 8, pitch, "verdict")
    print("# --- a second copy of the window one bank pair further (needs 2 x the LDS)")
    for rule in ("row", "col", "sum"):
        run(f"12 x 20 row-major, second copy for odd lag {rule}", 12, 20, 121, copy_rule=rule)
    run("16 x 16 row-major, second copy for odd lag row", 16, 16, 121, copy_rule="row")
    print("# --- second copy + a pitch that is a multiple of 32 elements (a window-row change then keeps the bank pair)")
    for sw, sh in ((16, 16), (12, 20), (32, 8)):
        for pitch in (128, 160):
            run(f"{sw} x {sh} row-major, pitch {pitch}, no copy", sw, sh, pitch)
            run(f"{sw} x {sh} row-major, pitch {pitch}, second copy for odd lag row", sw, sh, pitch, copy_rule="row")
    run("16 x 16 row-major, pitch 144 (16 mod 32), second copy for odd lag row", 16, 16, 144, copy_rule="row")
    print("# --- de-interleaved columns (even | odd halves of a row)")
    run("12 x 20 row-major, de-interleaved, pitch 121", 12, 20, 121, deinterleave=True)
    run("32 x 8 row-major, de-interleaved, pitch 129", 32, 8, 129, deinterleave=True)


if __name__ == "__main__":
    main()
