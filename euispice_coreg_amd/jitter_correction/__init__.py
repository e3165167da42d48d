from .jitter_correction import jitter_correction_imagers  # noqa: F401
