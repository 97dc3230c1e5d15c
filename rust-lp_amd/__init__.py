"""MI355X-native revised-simplex pivot engine behind RELP's trait surface.

Product code: HIP kernels + C-ABI (`csrc/`, `include/relp_engine.h`) and the host-side mirror of
the reference interface (`engine.py`).  The HIP extension is mandatory: there is no CPU fallback.
"""
from . import synthetic  # noqa: F401
from .matrix_data import MatrixData  # noqa: F401
from . import engine  # noqa: F401
