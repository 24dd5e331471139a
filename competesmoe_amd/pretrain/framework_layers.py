"""The three mixins the pretrain MoE classes inherit (part of the drop-in boundary):
how aux losses and logs leave the layer.  Same public API as
moe_pretrain_model/framework/layers/{regularized_layer.py:9-62, logging_layer.py:9-49, once_per_iter_layer.py:1-27}."""
from typing import Any, Callable, Dict

import torch


class RegularizedLayer:
    def __init__(self) -> None:
        super().__init__()
        self.reg_accumulated = {}
        self.reg_counts_n = {}
        self.regularization_present = False

    @property
    def reg_enabled(self) -> bool:
        return self.training and self.regularization_present

    def add_reg(self, loss_fn: Callable[[], torch.Tensor], name: str = "reg"):
        if self.reg_enabled:
            v = loss_fn()
            if name in self.reg_accumulated:
                self.reg_accumulated[name] = self.reg_accumulated[name] + v
                self.reg_counts_n[name] += 1
            else:
                self.reg_accumulated[name] = v
                self.reg_counts_n[name] = 1

    def get_reg_loss(self) -> Dict[str, torch.Tensor]:
        out = {n: self.reg_accumulated[n] / self.reg_counts_n[n] for n in self.reg_accumulated}
        self.reg_accumulated = {}
        self.reg_counts_n = {}
        return out


class LoggingLayer:
    def __init__(self) -> None:
        super().__init__()
        self._logs = {}
        self._log_counts = {}
        self._custom_reductions = {}

    def custom_reduction(self, name: str, reduction):
        self._custom_reductions[name] = reduction

    def log(self, name: str, value: Any, drop_old: bool = False):
        if torch.is_tensor(value):
            value = value.detach()
        drop_old = drop_old or not isinstance(value, (torch.Tensor, float, int))
        if name in self._custom_reductions:
            self._logs.setdefault(name, []).append(value)
        elif name not in self._logs or drop_old:
            self._logs[name] = value
            self._log_counts[name] = 1
        else:
            self._logs[name] = self._logs[name] + value
            self._log_counts[name] += 1

    def get_logs(self) -> Dict[str, Any]:
        res = {}
        for k, v in self._logs.items():
            if k in self._custom_reductions:
                res[k] = self._custom_reductions[k](v)
            elif isinstance(v, (torch.Tensor, int, float)):
                res[k] = v / self._log_counts[k]
            else:
                res[k] = v
        self._logs = {}
        self._log_counts = {}
        return res


class OncePerIterLayer:
    def pre_train_forward(self):
        pass

    def post_train_forward(self):
        pass

    def before_loss(self):
        pass
