"""f16_mpc_oop_py_amd -- MI355X-native batched F-16 simulation + MPC engine.

Host-side mirror of the reference's `F16` environment (env.py:29) over a leading batch
dimension; all arithmetic runs in hand-written HIP kernels behind the C-ABI of
include/f16_hip.h (libf16hip.so).  See DESIGN.md.
"""
from . import parameters  # noqa: F401
from .env import F16Batch  # noqa: F401
