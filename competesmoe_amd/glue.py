"""Schedule and loss glue around the layers: what the reference's trainers do TO the MoE layers each run / step,
as plain functions over `nn.Module` trees, so a harness (or the reference's own callback / task classes) can drive the native
layers exactly as it drives its own.  No GPU work here.

LLaVA stack (moe_model/train/llava_trainer.py:1025-1081, `LLaVACallback`):
  * once, before training: walk the vision tower, then the projector, in `.modules()` order; every layer with a `total_steps`
    attribute gets `set_total_steps(total_steps[, id_layer=, prob_flips_final=])` -- CompeteSMoE layers thread ONE dict through all
    of them (each call returns it with its own entry added) and count `id_layer` up across both sub-models;
  * after every optimiser step: `set_current_steps(global_step)` on every layer that has `current_steps`.
Pretrain stack (moe_pretrain_model/tasks/transformer_lm_mixin.py:256-267, framework/task/simple_task.py:386-389, 303-306,
framework/layers/regularized_layer.py:65-104):
  * at model build: for layer id, every MoE in it: `prob_flips_final` of the previous call is handed on, then
    `set_total_steps(id_layer=id)`;
  * before every step: `set_current_steps(iter)`;
  * after the forward: `LayerRegularizer.get(iter)` sums `get_reg_loss()` of every RegularizedLayer by name, scales, optionally
    decays linearly, and the task adds `reg_loss * args.reg` to the LM loss.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, Optional, Set, Tuple, Union

import torch
import torch.nn as nn


# ------------------------------------------------------------------------------------------------ LLaVA stack
def llava_on_train_begin(sub_models: Iterable[nn.Module], total_steps: int) -> Dict[int, torch.Tensor]:
    """`sub_models` in the reference's order: (vision_tower, mm_projector).  Returns the threaded `prob_flips_final` dict."""
    from .moe.moe import MoeLayer
    id_layer = 0
    layers: Dict[int, torch.Tensor] = {}
    for sub in sub_models:
        for module in sub.modules():
            if isinstance(module, MoeLayer) and hasattr(module, "total_steps"):
                if "competesmoe" in getattr(module.args, "moe_name", ""):
                    layers = module.set_total_steps(total_steps, id_layer=id_layer, prob_flips_final=layers)
                    id_layer += 1
                else:
                    module.set_total_steps(total_steps)
    return layers


def llava_on_step_end(sub_models: Iterable[nn.Module], global_step: int) -> None:
    from .moe.moe import MoeLayer
    for sub in sub_models:
        for module in sub.modules():
            if isinstance(module, MoeLayer) and hasattr(module, "current_steps"):
                module.set_current_steps(global_step)


# ------------------------------------------------------------------------------------------------ pretrain stack
def pretrain_init_schedules(layers: Iterable[nn.Module]):
    """`layers` = the transformer's encoder layers in order.  Returns the last `prob_flips_final`."""
    from .pretrain.moe import MoE
    prev = None
    for i, layer in enumerate(layers):
        for module in layer.modules():
            if isinstance(module, MoE) and hasattr(module, "total_steps"):
                if i > 0:
                    module.prob_flips_final = prev
                prev = module.set_total_steps(id_layer=i)
    return prev


def pretrain_set_step(model: nn.Module, it: int) -> None:
    from .pretrain.moe import MoE
    for module in model.modules():
        if isinstance(module, MoE) and hasattr(module, "current_steps"):
            module.set_current_steps(it)


class LayerRegularizer:
    """Sum of the layers' regularisation losses by name (same constructor / `get` contract as the reference class)."""

    def __init__(self, module: Union[nn.Module, Iterable[nn.Module]], stop_after: Optional[int] = None,
                 scales: Optional[Dict[str, float]] = None, lin_decay: Iterable[str] = (), options: Optional[Dict[str, Any]] = None):
        self.modules = []
        self.scales = dict(scales or {})
        self.stop_after = stop_after
        self.lin_decay: Set[str] = set(lin_decay)
        if self.lin_decay and stop_after is None:
            raise ValueError("Please specify stop_after when using lin_decay.")
        for m in ([module] if isinstance(module, nn.Module) else module):
            self.add_module(m)

    def add_module(self, module: nn.Module) -> None:
        from .pretrain.framework_layers import RegularizedLayer
        for name, m in module.named_modules():
            if isinstance(m, RegularizedLayer):
                self.modules.append((name, m))
                m.regularization_present = True

    def get(self, it: int) -> Tuple[Any, Dict[str, torch.Tensor]]:
        res: Dict[str, Any] = {}
        for _, m in self.modules:
            for k, v in m.get_reg_loss().items():
                res[k] = res.get(k, 0) + v
        to_log = {k: v.detach() for k, v in res.items()}
        for k in res:
            res[k] = res[k] * self.scales.get(k, 1)
        for k in self.lin_decay:
            if k in res:
                res[k] = res[k] * (1 - it / self.stop_after)
        return sum(res.values()), to_log


def total_loss(lm_loss: torch.Tensor, regularizer: LayerRegularizer, it: int, reg_scale: float = 1.0):
    """`res.loss + reg_loss * args.reg` (simple_task.py:303-306).  Returns (total, reg_log)."""
    reg_loss, reg_log = regularizer.get(it)
    return lm_loss + reg_loss * reg_scale, reg_log
