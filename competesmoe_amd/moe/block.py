"""The MoE half of the pre-LN encoder block around the layer:

    residual = hidden_states
    hidden_states = self.layer_norm2(hidden_states)
    results, auxiliary_loss, id_experts, infor_aux = self.moelayer(hidden_states, return_id_experts, is_vision=True)
    hidden_states = residual + results                      (siglip_smoe.py:152-155, SiglipEncoderMoELayer.forward)

as three fused steps instead of five passes over [T, D]:
  * LayerNorm and the router's gate projection in ONE launch (`csmoe_layernorm_gate`: the gate reads the normalised rows from LDS);
  * the residual add as the epilogue of the combine kernel (`csmoe_combine(..., residual)`);
  * in the backward, the residual-path gradient and the sum of the expert-path and gate-path gradients of LayerNorm's output
    are folded into the LayerNorm backward kernel (`csmoe_layernorm_bwd(dxn, dxn2, ..., add)`).
Rounding sequence is the reference's: xn rounded to x.dtype before the gate and the experts see it, `results` rounded before the
residual is added.  The attention half of the encoder layer is outside the path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..functional import LayerNormGate
from .moe import MoeLayer


class MoEBlock(nn.Module):
    """`hidden + moelayer(layer_norm(hidden))`.  `layer_norm` is the encoder layer's own nn.LayerNorm (its parameters stay where
    checkpoints expect them: `layer_norm2.{weight,bias}`), `moelayer` any layer of competesmoe_amd.moe.  Returns what the
    reference's encoder layer returns: (hidden_states, auxiliary_loss, id_experts, infor_aux)."""

    def __init__(self, layer_norm: nn.LayerNorm, moelayer: MoeLayer):
        super().__init__()
        if not isinstance(layer_norm, nn.LayerNorm) or len(layer_norm.normalized_shape) != 1:
            raise ValueError("MoEBlock: layer_norm must be an nn.LayerNorm over the last dimension")
        self.layer_norm2 = layer_norm
        self.moelayer = moelayer

    def forward(self, hidden_states: torch.Tensor, return_id_experts: bool = False):
        B, N, D = hidden_states.shape
        ln, layer = self.layer_norm2, self.moelayer
        xn, logits, xres = LayerNormGate.apply(hidden_states.reshape(B * N, D), ln.weight, ln.bias, ln.eps, layer.gate.weight)
        layer._pre_logits = logits
        # only layers whose output IS one combine can take the residual in its epilogue (smoe, competesmoe)
        layer._residual = xres if getattr(layer, "_fuses_residual", False) else None
        fused = layer._residual is not None
        try:
            results, auxiliary_loss, id_experts, infor_aux = layer(xn.view(B, N, D), return_id_experts, is_vision=True)
            if not fused or layer._residual is not None:      # shared-expert variants, or the combine did not take it
                results = xres.view(B, N, D) + results
        finally:
            layer._pre_logits = None
            layer._residual = None
        return results, auxiliary_loss, id_experts, infor_aux
