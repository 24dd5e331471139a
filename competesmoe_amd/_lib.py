"""ctypes binding of the C-ABI HIP library (include/csmoe.h).

The HIP library IS the product path: importing this module on a machine without the built
`lib/libcsmoe_hip.so` raises, and no CPU fallback exists anywhere in `competesmoe_amd`.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 and must be loaded FIRST, so that this library's DT_NEEDED entry binds
# to the runtime instance torch uses (same device context, streams and allocations).  Loaded the other way round the process holds
# two HIP runtimes and launches from this library fail with "no ROCm-capable device is detected" (seen with build() then smoke()
# in one process).
import torch  # noqa: F401  (load order, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSMOE_LIB") or os.path.join(_HERE, "lib", "libcsmoe_hip.so")   # CSMOE_LIB: A/B builds only

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_GELU_TANH, ACT_SILU, ACT_QUICK_GELU = 0, 1, 2, 3, 4, 5
SEL_SOFTMAX, SEL_RAW, SEL_TOPK_SOFTMAX, SEL_SIGMOID, SEL_TOPK_SIGMOID = 0, 1, 2, 3, 4
COMBINE_SEQ, COMBINE_DOT, COMBINE_SEQ_RW = 0, 1, 2
B_NK, B_KN = 0, 1
EPI_PLAIN, EPI_BIAS, EPI_BIAS_ACT, EPI_ACTGRAD, EPI_ROUND_BIAS32_ACT, EPI_SOFTPLUS_ROWSUM, EPI_SOFTPLUS_GRAD = 0, 1, 2, 3, 4, 5, 6
EPI_ACTGRAD_ROWSCALE = 7

ACT_CODES = {"none": ACT_NONE, "relu": ACT_RELU, "gelu": ACT_GELU, "gelu_tanh": ACT_GELU_TANH, "silu": ACT_SILU,
             "quick_gelu": ACT_QUICK_GELU}

_p, _i, _l = C.c_void_p, C.c_int, C.c_int64

# name -> (restype, argtypes); must list every symbol declared in include/csmoe.h
SIGNATURES = {
    "csmoe_version": (_i, []),
    "csmoe_last_error": (C.c_char_p, []),
    "csmoe_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "csmoe_gate_logits": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "csmoe_router_select": (_i, [_p, _i, _i, _i, _i, _i, _i, C.c_float, _p, _p, _p, _p]),
    "csmoe_router_select_bwd": (_i, [_p, _i, _i, _i, _i, _i, _i, C.c_float, _p, _p, _p, _p, _p, _p, _p]),
    "csmoe_bin_workspace_bytes": (_l, [_i, _i]),
    "csmoe_bin_tokens": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "csmoe_grouped_gemm_rowdot_cols": (_i, [_i, _i, _i, _l, _l, _l, _i]),
    "csmoe_affinity_finish": (_i, [_p, _i, _i, _i, _p, _l, _i, _p]),
    "csmoe_bin_tokens_hist": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "csmoe_gate_select_ok": (_i, [_i, _i, _i, _i, _i]),
    "csmoe_gate_select_rows": (_i, []),
    "csmoe_gate_select": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, C.c_float, _i, _p, _p, _p, _p, _p, _p]),
    "csmoe_dispatch_rows": (_i, [_p, _p, _i, _p, _i, _i, _i, _p]),
    "csmoe_dispatch_tokens": (_i, [_p, _p, _i, _p, _i, _i, _i, _p]),
    "csmoe_dispatch_rows_bwd": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _p, _p, _p]),
    "csmoe_combine": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "csmoe_combine_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "csmoe_grouped_gemm": (_i, [_p, _l, _p, _i, _l, _p, _p, _i, _i, _i, _i, _p, _p, _p, _l, _i, _i, _i, _i, _p]),
    "csmoe_dense_gemm": (_i, [_p, _l, _p, _i, _l, _p, _i, _i, _i, _p, _p, _p, _l, _i, _i, _i, _i, _p]),
    "csmoe_grouped_wgrad": (_i, [_p, _l, _p, _l, _p, _i, _i, _i, _i, _p, _l, _i, _i, _i, _i, _p, _p]),
    "csmoe_expert_order": (_i, [_p, _i, _p, _p]),
    "csmoe_chunk_offsets": (_i, [_p, _i, _i, _i, _p, _p]),
    "csmoe_dense_wgrad": (_i, [_p, _l, _p, _l, _i, _i, _i, _p, _l, _i, _i, _i, _i, _p]),
    "csmoe_grouped_colsum": (_i, [_p, _l, _p, _i, _i, _p, _i, _i, _p]),
    "csmoe_dense_colsum": (_i, [_p, _l, _i, _i, _p, _i, _i, _p]),
    "csmoe_softplus_mean": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "csmoe_softplus_mean_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "csmoe_router_aux_workspace_floats": (_l, [_i, _i, _i]),
    "csmoe_router_aux": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "csmoe_router_aux_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "csmoe_gate_bwd_small_ok": (_i, [_i, _i, _i]),
    "csmoe_gate_bwd_dx": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "csmoe_gate_bwd_dw_ranges": (_i, [_i, _i, _i]),
    "csmoe_gate_bwd_dw": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "csmoe_pair_cosine": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "csmoe_pair_cosine_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "csmoe_layernorm_gate": (_i, [_p, _p, _p, C.c_float, _p, _p, _p, _i, _i, _i, _p, _p, _i, _p]),
    "csmoe_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "csmoe_layernorm_bwd_blocks": (_i, [_i]),
    "csmoe_layernorm_gate_mixed": (_i, [_p, _p, _p, C.c_float, _p, _p, _p, _i, _i, _p, _p, _i, _p]),
    "csmoe_layernorm_bwd_mixed": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "csmoe_combine_mixed": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "csmoe_combine_bwd_mixed": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "csmoe_dispatch_rows_bwd_mixed": (_i, [_p, _p, _i, _p, _p, _i, _i, _p]),
    "csmoe_widen_sum": (_i, [_p, _p, _p, _p, _l, _p]),
    "csmoe_grouped_gemm_f32w": (_i, [_p, _l, _p, _l, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _l, _i, _i, _p]),
    "csmoe_quantize_mxfp8": (_i, [_p, _p, _i, _l, _i, _i, _i, _i, _p, _p, _p]),
    "csmoe_quantize_mxfp8_both": (_i, [_p, _p, _i, _l, _i, _i, _i, _p, _p, _p, _p, _p]),
    "csmoe_grouped_gemm_mxfp8": (_i, [_p, _l, _p, _l, _p, _p, _l, _l, _p, _p, _i, _i, _i, _i, _p, _p, _p, _l, _i, _i, _p]),
    "csmoe_dense_gemm_mxfp8": (_i, [_p, _l, _p, _l, _p, _p, _l, _l, _p, _i, _i, _i, _p, _p, _p, _l, _i, _i, _p]),
}


class CsmoeError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C competesmoe_amd/csrc`).  competesmoe_amd has no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib.csmoe_last_error().decode()
        if rc == 1:
            raise ValueError(f"csmoe {what}: {msg}")
        raise CsmoeError(f"csmoe {what}: {msg} (code {rc})")
