"""
euispice_coreg_amd -- MI355X-native (HIP / gfx950) implementation of the euispice_coreg
`hdrshift.Alignment` correlation sweep, behind the reference's own Python API:

    from euispice_coreg_amd.hdrshift import Alignment, AlignmentResults

The per-lag resample + Pearson work runs in libcoreg_hip.so (C ABI: include/coreg_hip.h).
There is no CPU fallback in this package.
"""
__version__ = "0.1.0"
