"""
springcraft_amd — MI355X-native elastic-network normal-mode engine.

Drop-in for the hot path of `springcraft` (GNM / ANM assembly + eigendecomposition): the same
public names (reference: springcraft/__init__.py:12-15), computed by hand-written HIP kernels
for gfx950 behind a C ABI (``include/springcraft_hip.h``).  There is no CPU fallback.
"""

__version__ = "0.3.0"  # tracks the reference API version it mirrors (springcraft 0.3.0)

from .anm import *  # noqa: F401,F403
from .atoms import AtomArray, read_pdb_ca  # noqa: F401
from .forcefield import *  # noqa: F401,F403
from .gnm import *  # noqa: F401,F403
from .interaction import *  # noqa: F401,F403
from . import nma  # noqa: F401
