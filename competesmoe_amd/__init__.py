"""MI355X-native (gfx950) sparse-MoE hot path for CompeteSMoE-style training.

HIP kernels behind a C ABI (include/csmoe.h) + the reference's nn.Module / registry surface on top.
Importing this package loads `lib/libcsmoe_hip.so`; there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is not built)

__all__ = ["_lib"]
