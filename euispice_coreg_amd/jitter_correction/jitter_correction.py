"""
`jitter_correction_imagers` -- drop-in for euispice_coreg.jitter_correction.jitter_correction_imagers
(jitter_correction/jitter_correction.py:14-174): the main real-world caller of the alignment sweep.  A time series of
imager FITS files is cut in overlapping sublists; every image of a sublist is cross-correlated (Carrington frame by
default) with the *corrected* first image of that sublist, and written out with its pointing keywords corrected.

Same arguments and same outputs as the reference.  What changes is how the work is scheduled:
  * one long-lived GPU context: the sublist's reference image is decoded, uploaded and resampled on the Carrington
    grid ONCE per sublist and stays resident for all its sweeps (the reference re-does it for every image);
  * the image to align goes up as the FITS float32 pixels it is stored as, thresholds are applied on the device;
  * a reader thread decodes the next image(s), `pipeline_depth` (default 2) driver threads -- each with its own
    library context and stream on the GPU -- upload and sweep, and a writer thread fits / writes the previous result:
    the GPU always has the next sweep queued behind the running one;
  * every GPU of the machine from the plain script, as the reference uses every core (its sweeps fan out over a
    process pool, jitter_correction.py:214-224 -> alignment.py:692-744): the images of a sublist -- independent units
    once their reference is fixed -- are dealt round-robin to the visible devices, each with its own `pipeline_depth`
    driver threads and library contexts and its own resident copy of the prepared reference; all outputs of a sublist
    are on disk before the next one starts (its reference is one of them, jitter_correction.py:106, :137-138).
    Threads only, never a fork.  `device=` or COREG_SINGLE_DEVICE=1 keep one GPU;
  * with torch.distributed initialised (one process per GPU) the images of a sublist are dealt round-robin to the
    ranks instead, each rank on its own device; ranks meet at a barrier between sublists.  No data-path collective.
`parallelism`, `cpu_count` are accepted and ignored (they size the reference's process pool).
"""
from __future__ import annotations

import os
import shutil
import threading
import warnings
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .. import _lib, parallel
from ..hdrshift.alignment import Alignment
from ..utils import fits_io


def build_sublists(n_files: int, sublist_length: int, overlap: int):
    """jitter_correction.py:91-98: index sublists after the first file (stride `sublist_length`, length
    `sublist_length + overlap`) and "before" it.  The reference anchors both families at index 0, so the second one is
    always the single list [0] (nothing to align); it is returned for completeness."""
    idx = np.arange(n_files)
    after_ref = idx[idx[0]:] if n_files else idx
    after = [after_ref[n:n + sublist_length + overlap] for n in range(0, len(after_ref), sublist_length)]
    before_ref = idx[idx[0]::-1] if n_files else idx
    before = [before_ref[n:n + sublist_length + overlap] for n in range(0, len(before_ref), sublist_length)]
    return after, before


def _time_tag(date_avg) -> str:
    """`Time(DATE-AVG).fits[11:19]` with ':' -> '_' (jitter_correction.py:117-118) from the ISO string itself."""
    return str(date_avg)[11:19].replace(":", "_")


def jitter_correction_imagers(list_files_input, path_files_output, lonlims=None, latlims=None, shape=None,
                              lag_crval1=np.arange(-5, 5, 0.1), lag_crval2=np.arange(-5, 5, 0.1),
                              lag_cdelt1=np.arange(0, 1, 1), lag_cdelt2=np.arange(0, 1, 1),
                              lag_crota=np.arange(0, 1, 1), sublist_length=10, overlap=1, window_files_input=-1,
                              method_carrington_reprojection="fa", unit_lag="arcsec", path_figures=None,
                              plot_all_figures=False, parallelism=True, cpu_count=None, small_fov_value_max=None,
                              small_fov_value_min=None, alignement_method="carrington", device=None, prefetch=2,
                              pipeline_depth=2):
    """See the module docstring.  Returns the list of (index_to_align, index_ref, AlignmentResults) this rank
    produced, in processing order (the reference returns None; the corrected files are the product)."""
    if overlap == 0:
        raise ValueError("number of overlapping images between sublists can not be equal to 0.")
    list_files_input = list(list_files_input)
    dates = [fits_io.read_header(p, window_files_input)["DATE-AVG"] for p in list_files_input]
    parameter_alignment = {"lag_crval1": lag_crval1, "lag_crval2": lag_crval2, "lag_cdelt1": lag_cdelt1,
                           "lag_cdelt2": lag_cdelt2, "lag_crota": lag_crota}
    kwargs_carrington = {"lonlims": lonlims, "latlims": latlims, "shape": shape}
    sublists_after, _sublists_before = build_sublists(len(list_files_input), sublist_length, overlap)
    rank, world = parallel.world_info()
    if path_figures is not None:
        os.makedirs(path_figures, exist_ok=True)
        if plot_all_figures:
            warnings.warn("plot_co_alignment figures are not produced by this implementation")
    os.makedirs(path_files_output, exist_ok=True)

    done = []
    depth = max(1, int(pipeline_depth))
    devices = session_devices(device, world)
    prefetch = max(1, int(prefetch), len(devices))
    reader = ThreadPoolExecutor(max_workers=prefetch)
    # one pool of `depth` driver threads per device; thread t of device d owns library context (d, slot t)
    drivers = [ThreadPoolExecutor(max_workers=depth) for _ in devices]
    writer = ThreadPoolExecutor(max_workers=max(1, min(4, len(devices))))
    slots = threading.local()
    slot_ids = [iter(range(depth)) for _ in devices]
    slot_lock = threading.Lock()

    def drive(k_dev, index_to_align, index_ref, path_reference, fut_image):
        if not hasattr(slots, "id"):
            with slot_lock:
                slots.id = next(slot_ids[k_dev])
        dev, slot0 = devices[k_dev]
        A, corr = _align_hrieuv_with_hrieuv(
            large_fov_fits_path=path_reference, large_fov_window=window_files_input,
            small_fov_path=list_files_input[index_to_align], window_to_align=window_files_input,
            date_to_align=_time_tag(dates[index_to_align]), parameter_alignment=parameter_alignment,
            cpu_count=cpu_count, do_plot_figure=plot_all_figures,
            method_carrington_reprojection=method_carrington_reprojection, reference_date=dates[index_ref],
            parallelism=parallelism, alignement_method=alignement_method, small_fov_value_max=small_fov_value_max,
            small_fov_value_min=small_fov_value_min, unit_lag=unit_lag, device=dev,
            _preloaded_small=fut_image.result(), _return_corr=True, _handle_slot=slot0 + slots.id, **kwargs_carrington)
        out_path = os.path.join(path_files_output, os.path.basename(list_files_input[index_to_align]))
        # sub-lag Gaussian fit + corrected FITS in a writer thread: the GPU is already on the next image
        figure_path = None
        if path_figures is not None:
            figure_path = os.path.join(path_figures, f"correlation_{_time_tag(dates[index_to_align])}_"
                                                     f"{_time_tag(dates[index_ref])}.pdf")
        return writer.submit(_finish, A, corr, window_files_input, out_path, figure_path)

    try:
        for ii, list_ in enumerate(sublists_after):
            index_ref = int(list_[0])
            path_reference = os.path.join(path_files_output, os.path.basename(list_files_input[index_ref]))
            if ii == 0 and rank == 0:
                shutil.copyfile(list_files_input[index_ref], path_reference)
            _barrier(world)  # the reference of this sublist is on disk for every rank
            mine = [int(i) for k, i in enumerate(list_[1:]) if k % world == rank]
            # bounded look-ahead: at most prefetch + depth images per device are in flight at any time
            inflight = []
            for k, index_to_align in enumerate(mine):
                while len(inflight) >= prefetch + depth * len(devices):
                    i0, f0 = inflight.pop(0)
                    done.append((i0, index_ref, f0.result().result()))
                img = reader.submit(fits_io.load_for_upload, list_files_input[index_to_align], window_files_input)
                k_dev = k % len(devices)
                inflight.append((index_to_align, drivers[k_dev].submit(drive, k_dev, index_to_align, index_ref,
                                                                       path_reference, img)))
            for i0, f0 in inflight:
                # surfaces exceptions; every output of this sublist is on disk before the next one starts
                done.append((i0, index_ref, f0.result().result()))
            _barrier(world)
    finally:
        reader.shutdown(wait=True)
        for d in drivers:
            d.shutdown(wait=True)
        writer.shutdown(wait=True)
    return done


def session_devices(device=None, world=1):
    """[(physical device, first context slot)] the session drives from this process: the one asked for (`device=`, a
    rank of a torch.distributed job, COREG_SINGLE_DEVICE=1), else every visible GPU.  COREG_VIRTUAL_DEVICES=k (tests on a
    one-GPU box) gives k logical devices: logical device L is physical device L mod n with its own block of context
    slots."""
    if device is not None:
        return [(device, 0)]
    if world > 1:
        return [(None, 0)]  # (hdrshift.Alignment takes the rank's current device)
    if os.environ.get("COREG_SINGLE_DEVICE", "0") == "1":
        return [(-1, 0)]
    n_logical, n_physical = _lib.device_count(), _lib.physical_device_count()
    if n_logical <= 1 or n_physical < 1:
        return [(-1, 0)]
    return [(L % n_physical, 16 * (L // n_physical)) for L in range(n_logical)]


def _finish(A, corr, window, out_path, figure_path=None):
    results = A._wrap(corr, "AlignmentResults", restore_units=True)
    results.write_corrected_fits(window_list_to_apply_shift=[window], path_to_l3_output=out_path)
    if figure_path is not None:  # jitter_correction.py:241-243
        fig, _ = results.plot_correlation(path_save_figure=figure_path)
        from matplotlib import pyplot as plt
        plt.close(fig)
    return results


def _barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def _align_hrieuv_with_hrieuv(large_fov_fits_path, large_fov_window, small_fov_path, parameter_alignment,
                              date_to_align, cpu_count=30, window_to_align=3, do_plot_figure=False, parallelism=True,
                              lonlims=None, latlims=None, shape=None, unit_lag="arcsec", reference_date=None,
                              small_fov_value_max=None, small_fov_value_min=None, method_carrington_reprojection="fa",
                              alignement_method="carrington", path_output_figures=None, fov_limits=None, device=None,
                              _preloaded_small=None, _return_corr=False, _handle_slot=0):
    """jitter_correction.py:177-256: one image against the sublist's reference.  `_return_corr`: hand back
    (Alignment, raw correlation array) so that the caller can build the AlignmentResults off the critical path."""
    A = Alignment(large_fov_known_pointing=large_fov_fits_path, large_fov_window=large_fov_window,
                  small_fov_to_correct=small_fov_path, small_fov_window=window_to_align, display_progress_bar=False,
                  small_fov_value_max=small_fov_value_max, small_fov_value_min=small_fov_value_min,
                  parallelism=parallelism, counts_cpu_max=cpu_count, unit_lag=unit_lag, device=device,
                  **parameter_alignment)
    A.shard_lags = False
    A._preloaded_small = _preloaded_small
    A._handle_slot = _handle_slot
    rt = "corr" if _return_corr else "AlignmentResults"
    if alignement_method == "carrington":
        res = A.align_using_carrington(method="correlation", lonlims=lonlims, latlims=latlims, shape=shape,
                                       reference_date=reference_date, return_type=rt,
                                       method_carrington_reprojection=method_carrington_reprojection)
    elif alignement_method == "initial_carrington":
        res = A.align_using_initial_carrington(method="correlation", return_type=rt)
    elif alignement_method == "helioprojective":
        res = A.align_using_helioprojective(method="correlation", fov_limits=fov_limits, return_type=rt)
    else:
        raise ValueError("alignement_method must be 'carrington', 'initial_carrington' or 'helioprojective'")
    return (A, res) if _return_corr else res
