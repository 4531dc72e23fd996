"""MI355X-native RGB-D front end for ManhattanSLAM: ORB extractor + surfel fusion.

The product is the C-ABI shared library ``libmsl.so`` (hand-written HIP for gfx950, see
``include/msl.h``); this package is the thin Python host mirror of the reference's two
entry classes, used by the tests and ``bench.py``.  There is no CPU fallback: importing the
compute classes without the built library, or constructing them without an MI355X, raises.
"""
from ._lib import lib, MslError, KEYPOINT_DTYPE, SURFEL_DTYPE, SEED_DTYPE, device_count  # noqa: F401
from .orb import ORBextractor, frame_params  # noqa: F401
from .surfel import SurfelFusion, SurfelMap  # noqa: F401
from . import peac  # noqa: F401
from ._lib import PEAC_STATS_DTYPE  # noqa: F401
from . import match  # noqa: F401
from ._lib import MATCH_PARAMS_DTYPE  # noqa: F401
