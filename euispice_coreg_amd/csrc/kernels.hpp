// HIP kernels (gfx950 / CDNA4, wave64) of the alignment sweep.
//
// Work decomposition ("lag-per-lane"):
//   * the target grid is cut in tiles of 1024 points; a precompute pass (k_precompute) evaluates the
//     lag-independent part of the pixel-coordinate transform once per tile, drops points that can never
//     contribute (reference NaN, behind the limb, outside the small image for every lag) and stores the
//     survivors compacted, tile-major, together with their bounding box in small-image pixel space;
//   * the sweep kernel (k_sweep) gives every LANE one lag-point (a workgroup = 4 point-groups x 256 lags of a compact
//     CRVAL patch, all sharing one LDS window) and walks the compacted points of its tiles: point data are wave-uniform (scalar loads), every lane adds
//     its own lag displacement, gathers the 3x3 (order 2) / 2x2 (order 1) taps from an LDS-staged window of the
//     small image and accumulates its own six Pearson sums in registers.  No cross-lane reduction exists
//     anywhere on the hot path; partial sums leave the kernel once per (tile-group, lag);
//   * k_finalize adds the tile-group slabs in a fixed order (deterministic) and evaluates the coefficient.
//
// Reference arithmetic restated (paths relative to euispice_coreg/):
//   utils/Util.py:82-104 + scipy.ndimage.map_coordinates(order, mode='constant', prefilter=False)  -> spline_*()
//   utils/rectify.py:340-363 SphericalTransform.forward                                             -> carr_term()
//   hdrshift/alignment.py:525-531 mask + hdrshift/c_correlate.py:39-72 Pearson                      -> k_sweep/k_finalize
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels_common.hpp"    // constants, argument structs, per-lag maps, spline weights
#include "kernels_prepare.hpp"   // pivots, FITS decode, resample, k_precompute, k_tile_list
#include "kernels_sweep.hpp"     // Taps, gathers, point_lag, tile_points, k_sweep
#include "kernels_fix.hpp"       // noise-decided samples: border / parity / tap scan + fix
#include "kernels_finalize.hpp"  // k_finalize, k_refine_list, k_refine
