"""Pretrain `smoe` with expert-parallel experts (`smoe_ep`).

The reference trains data-parallel only; this is the pretrain stack's counterpart of `competesmoe_amd.ep.EPSMoeLayer`
(DESIGN.md section 5), with one contract: the numbers of the single-GPU `smoe` layer (moe_pretrain_model/layers/moe/smoe.py:38-263) on
the same tokens.  `n_experts` is the GLOBAL expert count; `keys` / `values` (/ `bias`) hold THIS rank's E/P experts (global ids
rank*E/P ...), so a state dict is the rank's slice of the packed tensors.  The gate (and `o_bias`) are replicated: their gradients are
summed over the group during backward (tokens are data-parallel); expert gradients are local by construction.  The entropy-balance
regulariser is computed per rank on local tokens, as data-parallel training of the reference computes it."""
import os
from typing import Optional

import torch
import torch.distributed as dist

from .. import _lib as L
from ..ep import EPFFNPacked, _direct_default, reduce_grad_on_backward
from .moe import op_dtype
from .register import register_moe
from .smoe import SMoeLayer


@register_moe("smoe_ep")
class EPSMoeLayer(SMoeLayer):
    def __init__(self, *args, group=None, chunks: Optional[int] = None, direct: Optional[bool] = None, **kwargs):
        if not dist.is_initialized():
            raise RuntimeError("competesmoe_amd: smoe_ep needs an initialised torch.distributed process group")
        object.__setattr__(self, "group", group)        # read by _n_held_experts inside the constructor
        super().__init__(*args, **kwargs)
        if self.is_att:
            raise ValueError("smoe_ep: the MoE-attention projections are not expert-parallel")
        if self.fp8_experts:
            raise ValueError("smoe_ep: args.fp8_experts is a single-GPU path")
        self.chunks = chunks        # groups of local experts whose exchanges overlap the GEMMs; None: CSMOE_EP_CHUNKS, else 2 (1 at P=1)
        self.direct = direct        # one message per (peer, local expert), no regroup passes; None: CSMOE_EP_DIRECT, else off
        reduce_grad_on_backward(self.w_gate, group)
        if self.o_bias is not None:
            reduce_grad_on_backward(self.o_bias, group)

    def _n_held_experts(self, n_experts: int) -> int:
        P = dist.get_world_size(self.group)
        if n_experts % P:
            raise ValueError(f"smoe_ep: {n_experts} experts do not divide over {P} ranks")
        return n_experts // P

    def _n_chunks(self) -> int:
        c = self.chunks
        if c is None:
            env = os.environ.get("CSMOE_EP_CHUNKS")
            c = int(env) if env else (2 if dist.get_world_size(self.group) > 1 else 1)
        return max(1, min(int(c), self.keys.shape[0]))

    def ffn(self, x, selected_experts, weights, keys=None, values=None, bias=None):
        if keys is not None or values is not None or bias is not None:
            raise ValueError("smoe_ep: the experts are the layer's own sharded tensors")
        shp = x.shape
        x2 = self.operand(x)
        K = selected_experts.shape[-1]
        res = None
        if self._residual is not None:
            res, self._residual = self._residual, None
        stats = {} if (self.log_interval is not None and self.iter % self.log_interval == 0) else None
        direct = _direct_default() if self.direct is None else bool(self.direct)
        out = EPFFNPacked.apply(x2, weights.reshape(-1, K).float().contiguous(), selected_experts.reshape(-1, K).int().contiguous(),
                                self.keys, self.values, self.bias, None, self.act_code, L.COMBINE_DOT, self.n_experts, self.group,
                                self._n_chunks(), direct, res, stats)
        if stats:
            with torch.no_grad():
                h = stats["hact"]
                if h.numel():
                    self.log("relu_pass_rate", (h > 0).float().sum() / h.numel())
        return out.view(*shp[:-1], -1)
