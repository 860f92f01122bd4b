"""algames.jl_amd -- MI355X-native batched ALGAMES Newton / augmented-Lagrangian hot path.

Import as `algames_jl_amd` (see algames_jl_amd.py at the repo root).  `host` mirrors the reference's
Julia API for this path; `_abi` is the ctypes mirror of include/algames_hip.h; `csrc/` holds the HIP
kernels and the C ABI; `scenarios` generates the synthetic BASELINE configurations.
"""
from ._abi import *  # noqa: F401,F403
from ._abi import Batch, CLib, AlgamesError  # noqa: F401
from .host import *  # noqa: F401,F403
from .host import hip_lib, HIP_LIB_PATH  # noqa: F401
from . import scenarios  # noqa: F401,E402
from . import active_set  # noqa: F401,E402
from . import sharding  # noqa: F401,E402
from .sharding import ShardedGameProblem  # noqa: F401,E402
