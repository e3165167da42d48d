from .alignment import Alignment  # noqa: F401
from .alignment_results import AlignmentResults  # noqa: F401
from .alignment_spice import AlignmentSpice  # noqa: F401
