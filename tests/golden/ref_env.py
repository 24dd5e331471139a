"""Build-container-only environment in which the reference's pretrain stack RUNS ON CPU with its OWN Triton kernels.

Used by make_golden_pretrain.py / make_golden_pretrain_block.py.  Nothing here restates reference arithmetic; it repairs and
configures the TOOLS the reference runs on:

1. Triton interpreter (`TRITON_INTERPRET=1`, set before triton is imported): `layers/cvmm.py`'s `cvmm_kernel` and
   `cvmm_backward_kernel3` execute op by op on numpy.  Two triton-3.6 interpreter defects are patched (tool, not reference):
   * `InterpreterBuilder.create_dot` runs `np.matmul` on the uint16 STORAGE of bf16 operands (garbage).  Patched: bf16 operands
     are widened to fp32 first (exact), the product accumulates in fp32 -- what an MFMA / tensor-core bf16 dot does.
   * `_convert_float` fp32 -> bf16 truncates (cast path) / mis-rounds on a mantissa carry.  Patched to round-to-nearest-even,
     the conversion `x.to(tl.bfloat16)` compiles to on a GPU.
2. The autotuner needs a GPU driver to benchmark its candidates: each autotuned kernel is replaced by a launcher that calls the
   SAME jitted function (`Autotuner.fn`) with ONE configuration taken from the kernel's own candidate list (recorded in the
   fixtures' meta).  `cvmm_triton_backward` zero-fills its output itself, so `reset_to_zero` is not needed.
3. `cvmm()` asks `torch.cuda.get_device_properties(0)` for the Volta check (cvmm.py:557-560): answered with a non-Volta
   capability.
4. The reference trains under `torch.cuda.amp.autocast(bf16)` (simple_task.py:295).  `cuda_autocast_bf16()` reproduces that on
   CPU tensors: CPU autocast supplies the lower-precision casts (linear / matmul / bmm -> bf16, same as the CUDA list for the ops
   on this path), the thread-local CUDA autocast flags make `cvmm.get_dtype()` return bf16, and a TorchFunctionMode applies the
   CUDA autocast fp32 policy (aten/src/ATen/autocast_mode.cpp: softplus, softmax, log_softmax, sum, norm, exp, log, pow,
   layer_norm, logsumexp, mse_loss, ...) that CPU autocast does not have.  Every function the mode upcast is counted and
   stored in the fixtures' meta.
"""
import contextlib
import importlib
import importlib.util
import os
import sys
import types

os.environ["TRITON_INTERPRET"] = "1"

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.overrides import TorchFunctionMode  # noqa: E402

REF = "/root/reference/moe_pretrain_model"

# one candidate of each kernel's own autotune list (cvmm.py:56 and :173)
FWD_CONFIG = dict(BLOCK_SIZE_M=32, BLOCK_SIZE_N=64, BLOCK_SIZE_K=32, GROUP_SIZE_M=8)
BWD_CONFIG = dict(BLOCK_SIZE_M=64, BLOCK_SIZE_N=64, BLOCK_SIZE_K=16, GROUP_SIZE_M=8, K_BLOCKS=64)
CVMM_META = ("reference triton kernels (cvmm_kernel, cvmm_backward_kernel3) under TRITON_INTERPRET=1; "
             f"fixed configs fwd={FWD_CONFIG} bwd={BWD_CONFIG}; interpreter bf16 dot / RTNE cast repaired (tests/golden/ref_env.py)")


# ------------------------------------------------------------------------------------------------ triton interpreter repairs
def _patch_interpreter():
    import triton.language as tl
    from triton.runtime import interpreter as ti

    if getattr(ti, "_csmoe_patched", False):
        return
    orig_convert = ti._convert_float

    def convert_float(inp, in_dt, out_dt, rounding_mode):
        if in_dt == tl.float32 and out_dt == tl.bfloat16:
            a = np.ascontiguousarray(inp).view(np.float32)
            t = torch.from_numpy(a.copy()).to(torch.bfloat16).view(torch.int16)
            return t.numpy().view(np.uint16).reshape(a.shape)
        if in_dt == tl.bfloat16 and out_dt == tl.float32:
            a = np.ascontiguousarray(inp).view(np.uint16)
            return (a.astype(np.uint32) << 16).view(np.uint32).reshape(a.shape)
        return orig_convert(inp, in_dt, out_dt, rounding_mode)

    ti._convert_float = convert_float

    def widen(h):
        if h.dtype.scalar == tl.bfloat16:
            return (np.ascontiguousarray(h.data).view(np.uint16).astype(np.uint32) << 16).view(np.float32)
        return h.data

    def create_dot(self, a, b, d, input_precision, max_num_imprecise_acc):
        assert not (a.dtype.primitive_bitwidth == 8 or b.dtype.primitive_bitwidth == 8)
        return ti.TensorHandle(np.matmul(widen(a), widen(b), dtype=d.data.dtype) + d.data, d.dtype.scalar)

    ti.InterpreterBuilder.create_dot = create_dot
    ti._csmoe_patched = True


class _FixedConfig:
    """`kernel[grid](*args, **kw)` for an autotuned kernel, with one configuration and no benchmarking."""

    def __init__(self, autotuner, config):
        self.fn, self.config = autotuner.fn, dict(config)

    def __getitem__(self, grid):
        def launch(*a, **kw):
            g = grid(self.config) if callable(grid) else grid
            return self.fn[g](*a, **kw, **self.config)
        return launch


# ------------------------------------------------------------------------------------------------ reference import
def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def import_reference():
    """Stub PACKAGES (the pretrain package does not import as shipped) whose __path__ points into the
    reference; the mixins, entropy / distributed_ops helpers and layers/cvmm.py loaded by file path; layers.moe.* imported
    normally.  Returns the reference's get_moe."""
    if "layers.moe.smoe" in sys.modules:
        return sys.modules["layers.moe.register"].get_moe
    _patch_interpreter()
    torch.cuda.get_device_properties = lambda *a, **k: types.SimpleNamespace(major=9, minor=4, name="cpu-interpreter")
    fw = _pkg("framework", os.path.join(REF, "framework"))
    fwl = _pkg("framework.layers", os.path.join(REF, "framework", "layers"))
    fwu = _pkg("framework.utils", os.path.join(REF, "framework", "utils"))

    # minimal stand-in for framework.utils.U (only apply_to_tensors is used by LoggingLayer.log)
    U = types.ModuleType("framework.utils.U")

    def apply_to_tensors(d, fn):
        if torch.is_tensor(d):
            return fn(d)
        if isinstance(d, (list, tuple)):
            return type(d)(apply_to_tensors(v, fn) for v in d)
        if isinstance(d, dict):
            return {k: apply_to_tensors(v, fn) for k, v in d.items()}
        return d
    U.apply_to_tensors = apply_to_tensors
    fwu.U = U
    ent = _load("framework.utils.entropy", os.path.join(REF, "framework", "utils", "entropy.py"))
    dops = _load("framework.utils.distributed_ops", os.path.join(REF, "framework", "utils", "distributed_ops.py"))
    for k in ("entropy", "entropy_l", "relative_perplexity", "relative_perplexity_l", "perplexity"):
        setattr(fwu, k, getattr(ent, k))
    fwu.distributed_ops = dops
    fwu.entropy = ent.entropy
    fw.utils = fwu
    ll = _load("framework.layers.logging_layer", os.path.join(REF, "framework", "layers", "logging_layer.py"))
    rl = _load("framework.layers.regularized_layer", os.path.join(REF, "framework", "layers", "regularized_layer.py"))
    ol = _load("framework.layers.once_per_iter_layer", os.path.join(REF, "framework", "layers", "once_per_iter_layer.py"))
    fwl.LoggingLayer, fwl.RegularizedLayer, fwl.OncePerIterLayer = ll.LoggingLayer, rl.RegularizedLayer, ol.OncePerIterLayer
    fw.layers = fwl

    lay = _pkg("layers", os.path.join(REF, "layers"))
    cv = _load("layers.cvmm", os.path.join(REF, "layers", "cvmm.py"))
    cv.cvmm_kernel = _FixedConfig(cv.cvmm_kernel, FWD_CONFIG)
    cv.cvmm_backward_kernel3 = _FixedConfig(cv.cvmm_backward_kernel3, BWD_CONFIG)
    cv.print = lambda *a, **k: None            # "New shape: ..." chatter of cvmm()
    lay.cvmm = cv.cvmm
    lay.cvmm_prepare_sel = cv.cvmm_prepare_sel
    _pkg("layers.moe", os.path.join(REF, "layers", "moe"))
    for m in ("register", "moe", "smoe", "competesmoe", "deepseekv2", "deepseekv3", "smoe_perturbed"):
        importlib.import_module(f"layers.moe.{m}")
    return sys.modules["layers.moe.register"].get_moe


# ------------------------------------------------------------------------------------------------ CUDA autocast on CPU tensors
# aten/src/ATen/autocast_mode.cpp, CUDA tables: `fp32` and `fp32_set_opt_dtype` / `fp32_append_dtype` policies (the ops a
# transformer + MoE layer can reach); names as they appear as torch.* / Tensor.* / F.* callables.
_FP32_OPS = {
    "acos", "asin", "cosh", "erfinv", "exp", "expm1", "log", "log10", "log2", "log1p", "reciprocal", "rsqrt", "sinh", "tan",
    "pow", "__pow__", "__rpow__", "softplus", "layer_norm", "group_norm", "frobenius_norm", "nuclear_norm", "cosine_similarity",
    "poisson_nll_loss", "cosine_embedding_loss", "nll_loss", "hinge_embedding_loss", "kl_div", "l1_loss", "smooth_l1_loss",
    "huber_loss", "mse_loss", "margin_ranking_loss", "multilabel_margin_loss", "soft_margin_loss", "triplet_margin_loss",
    "multi_margin_loss", "binary_cross_entropy_with_logits", "dist", "pdist", "cdist", "renorm", "logsumexp",
    "prod", "softmax", "log_softmax", "cumprod", "cumsum", "sum", "norm", "vector_norm", "matrix_norm", "linalg_vector_norm",
    "cross_entropy",
}


def _tree(x, fn):
    if isinstance(x, torch.Tensor):
        return fn(x)
    if isinstance(x, (list, tuple)):
        return type(x)(_tree(v, fn) for v in x)
    if isinstance(x, dict):
        return {k: _tree(v, fn) for k, v in x.items()}
    return x


class CudaAutocastFp32Policy(TorchFunctionMode):
    def __init__(self):
        super().__init__()
        self.upcast = {}

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name == "normalize" and args and isinstance(args[0], torch.Tensor) and args[0].dtype == torch.bfloat16:
            # F.normalize = v / v.norm(p, dim, keepdim=True).clamp_min(eps).expand_as(v) (torch/nn/functional.py); under CUDA
            # autocast `norm` is an fp32 op, so the denominator is fp32 and the bf16 / fp32 division promotes to fp32
            self.upcast["normalize"] = self.upcast.get("normalize", 0) + 1
            v = args[0]
            p = kwargs.get("p", args[1] if len(args) > 1 else 2.0)
            dim = kwargs.get("dim", args[2] if len(args) > 2 else 1)
            eps = kwargs.get("eps", args[3] if len(args) > 3 else 1e-12)
            denom = v.float().norm(p, dim, keepdim=True).clamp_min(eps).expand_as(v)
            return v / denom
        if name in _FP32_OPS:
            hit = []

            def up(t):
                if t.dtype == torch.bfloat16:
                    hit.append(1)
                    return t.float()
                return t
            args, kwargs = _tree(args, up), _tree(kwargs, up)
            if hit:
                self.upcast[name] = self.upcast.get(name, 0) + 1
        return func(*args, **kwargs)


@contextlib.contextmanager
def cuda_autocast_bf16(log=None):
    """`with torch.cuda.amp.autocast(dtype=bf16)` for CPU tensors (see the module docstring, item 4)."""
    mode = CudaAutocastFp32Policy()
    was = torch.is_autocast_enabled("cuda")
    was_dt = torch.get_autocast_dtype("cuda")
    torch.set_autocast_enabled("cuda", True)
    torch.set_autocast_dtype("cuda", torch.bfloat16)
    try:
        with torch.autocast("cpu", dtype=torch.bfloat16), mode:
            yield mode
    finally:
        torch.set_autocast_enabled("cuda", was)
        torch.set_autocast_dtype("cuda", was_dt)
        if log is not None:
            for k, v in mode.upcast.items():
                log[k] = log.get(k, 0) + v
