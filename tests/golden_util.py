"""Helpers shared by the parity tests: load golden fixtures, unpack expert weights."""
import os
from types import SimpleNamespace

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ACT_OF_KIND = {"seq_gelu": "gelu", "siglip_tanh": "gelu_tanh"}


def load(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=True)


def expert_keys(kind, i):
    if kind == "seq_gelu":
        return (f"experts.{i}.0.weight", f"experts.{i}.0.bias", f"experts.{i}.2.weight", f"experts.{i}.2.bias")
    return (f"experts.{i}.fc1.weight", f"experts.{i}.fc1.bias", f"experts.{i}.fc2.weight", f"experts.{i}.fc2.bias")


def unpack_experts(fx, requires_grad=False):
    kind = fx["meta"]["expert_kind"]
    E = fx["meta"]["E"]
    out = []
    for i in range(E):
        ts = tuple(fx["state"][k].clone().requires_grad_(requires_grad) for k in expert_keys(kind, i))
        out.append(ts)
    return out


def args_of(fx):
    return SimpleNamespace(**fx["meta"]["args"])


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
