"""The MoE half of the pretrain stack's pre-LN transformer layer around the MoE layer:

    mlp_input = src
    src2 = self.norm2(mlp_input)
    src3 = self.pkm(src2, id_layer=id_layer)
    src = src + self.dropout(src3)              (relative_moe_transformer.py:153-161, RelativeMoeTransformerEncoderLayer.forward, preln)

as three fused steps instead of a dozen passes over [T, D]:
  * LayerNorm and the router's gate projection in one call (`csmoe_layernorm_gate[_mixed]`);
  * the residual add as the epilogue of the combine kernel;
  * in the backward, the residual-path gradient and the two gradient streams of LayerNorm's output folded into the LayerNorm
    backward kernel.
The training loop runs this under bf16 autocast (simple_task.py:295) on an fp32 residual stream: LayerNorm is an fp32 op there, the
gate's F.linear and cvmm cast its output to bf16, the bf16 MoE output is added to the fp32 stream in fp32, and autograd inserts the
mirror-image casts.  The mixed-precision kernels read and write the fp32 stream directly and keep every rounding where the
reference has it (xn rounded once to bf16; the combine result rounded to bf16 before the fp32 add; the upstream gradient rounded
to bf16 on load; the two bf16 gradients of xn summed in fp32).  Outside autocast (fp32, or a bf16 model) the same-dtype kernels of
the LLaVA block are used.  The attention half of the layer is outside the path."""
from __future__ import annotations

from typing import Optional, Union

import torch
import torch.nn as nn

from ..functional import LayerNormGate
from .moe import MoE, op_dtype


class MoEBlock(nn.Module):
    """`src + dropout(pkm(norm2(src), id_layer=id_layer))`.  `norm` is the transformer layer's own nn.LayerNorm (its parameters
    stay where checkpoints expect them: `norm2.{weight,bias}`), `pkm` any layer of competesmoe_amd.pretrain, `dropout` the layer's
    nn.Dropout (or its probability).  Returns the new residual stream, as the reference's layer does."""

    def __init__(self, norm: nn.LayerNorm, pkm: MoE, dropout: Union[nn.Dropout, float, None] = None):
        super().__init__()
        if not isinstance(norm, nn.LayerNorm) or len(norm.normalized_shape) != 1:
            raise ValueError("MoEBlock: norm must be an nn.LayerNorm over the last dimension")
        self.norm2 = norm
        self.pkm = pkm
        self.dropout = dropout if isinstance(dropout, nn.Module) else nn.Dropout(float(dropout or 0.0))

    def _fusable(self, src: torch.Tensor) -> bool:
        D = src.shape[-1]
        op = op_dtype(src)
        return (src.is_cuda and src.dim() == 3 and D % 8 == 0 and D <= 4096 and self.pkm._plain_gate()
                and (src.dtype == op or (src.dtype == torch.float32 and op == torch.bfloat16))
                and all(p is None or p.dtype == src.dtype for p in (self.norm2.weight, self.norm2.bias)))

    def forward(self, src: torch.Tensor, id_layer: Optional[int] = None) -> torch.Tensor:
        ln, layer = self.norm2, self.pkm
        if not self._fusable(src):
            return src + self.dropout(layer(ln(src), id_layer=id_layer))
        B, N, D = src.shape
        op = op_dtype(src)
        xn, logits, xres = LayerNormGate.apply(src.reshape(B * N, D), ln.weight, ln.bias, ln.eps, layer.w_gate, op)
        drop = self.training and self.dropout.p > 0
        fuse_res = (layer._fuses_residual and not drop and layer.o_bias is None and layer.v_dim == D)
        layer._pre_logits = logits
        layer._residual = xres if fuse_res else None
        layer._stream_dtype = src.dtype
        try:
            out = layer(xn.view(B, N, D), id_layer=id_layer)
            if not fuse_res or layer._residual is not None:      # shared-expert variants / dropout / the combine did not take it
                out = xres.view(B, N, D) + self.dropout(out)
        finally:
            layer._pre_logits = None
            layer._residual = None
            layer._stream_dtype = None
        return out
