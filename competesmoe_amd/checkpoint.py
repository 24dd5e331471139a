"""Checkpoint bridge between the two expert-weight formats of the reference.

LLaVA stack: per-expert modules, `experts.{i}.<fc1>.{weight [F,D], bias [F]}`, `experts.{i}.<fc2>.{weight [Dout,F], bias [Dout]}`
(`fc1/fc2` for SiglipMLP / CLIPMLP, `0/2` for the `nn.Sequential(Linear, act, Linear)` experts), `gate.weight [E,D]`
(moe_model/model/moe/moe.py:36-46).  Pretrain stack: packed parameters `keys [E,D,F]`, `values [E,F,Dout]`, optional `bias [E,F]`,
`o_bias [Dout]`, `w_gate [E,D]` (moe_pretrain_model/layers/moe/moe.py:120-134).  keys[e] = fc1.weight^T, values[e] = fc2.weight^T.

Not every layer converts: the pretrain layer has ONE output bias added after the weighted sum, the LLaVA experts have one per
expert that is scaled by the routing weight, so per-expert fc2 biases only pack when they are all zero / absent (anything else
raises).  Host-side tensor surgery only; nothing here touches the GPU path.

Also the two upcycling steps of moe_model/model/llava_arch.py: every expert initialised from one dense MLP (:110-119) and the key
surgery that maps `...moelayer.experts.{i}.<sub>.<param>` onto the dense checkpoint's `...mlp.<sub>.<param>` (:131-143).
"""
from __future__ import annotations

import re
from typing import Dict, List, Mapping, Optional, Tuple

import torch

_EXPERT_KEY = re.compile(r"^(?P<pre>.*?)experts\.(?P<i>\d+)\.(?P<sub>[^.]+)\.(?P<param>weight|bias)$")


def _split_expert_keys(sd: Mapping[str, torch.Tensor], prefix: str = "") -> Tuple[Dict[int, Dict[str, torch.Tensor]], List[str]]:
    """{expert index: {"<sub>.<param>": tensor}} for the keys under `prefix`, and the sub-module names in first-seen order."""
    per: Dict[int, Dict[str, torch.Tensor]] = {}
    subs: List[str] = []
    for k, v in sd.items():
        if not k.startswith(prefix):
            continue
        m = _EXPERT_KEY.match(k[len(prefix):])
        if not m or m.group("pre"):
            continue
        per.setdefault(int(m.group("i")), {})[f"{m.group('sub')}.{m.group('param')}"] = v
        if m.group("sub") not in subs:
            subs.append(m.group("sub"))
    return per, subs


def pack_llava_experts(sd: Mapping[str, torch.Tensor], prefix: str = "") -> Dict[str, torch.Tensor]:
    """LLaVA-format state dict of one MoE layer (keys `<prefix>gate.weight`, `<prefix>experts.{i}...`) -> pretrain-format
    tensors {w_gate, keys, values[, bias]}.  The two Linear sub-modules are recognised by order (fc1 / 0 first, fc2 / 2 second)."""
    per, subs = _split_expert_keys(sd, prefix)
    if not per:
        raise KeyError(f"no '{prefix}experts.<i>.<sub>.weight' keys found")
    if sorted(per) != list(range(len(per))):
        raise ValueError(f"expert indices are not 0..E-1: {sorted(per)}")
    lin = [s for s in subs if f"{s}.weight" in per[0]]
    if len(lin) != 2:
        raise ValueError(f"experts must be two-matrix FFNs, found Linear sub-modules {lin}")
    s1, s2 = lin
    E = len(per)
    w1 = [per[e][f"{s1}.weight"] for e in range(E)]
    w2 = [per[e][f"{s2}.weight"] for e in range(E)]
    F_, D = w1[0].shape
    if any(w.shape != (F_, D) for w in w1) or any(w.shape[1] != F_ for w in w2):
        raise ValueError("experts have different shapes")
    out = {"keys": torch.stack([w.t() for w in w1]).contiguous(),        # [E, D, F]
           "values": torch.stack([w.t() for w in w2]).contiguous()}      # [E, F, Dout]
    if f"{s1}.bias" in per[0]:
        out["bias"] = torch.stack([per[e][f"{s1}.bias"] for e in range(E)]).contiguous()
    if f"{s2}.bias" in per[0]:
        b2 = torch.stack([per[e][f"{s2}.bias"] for e in range(E)])
        if bool((b2 != 0).any()):
            raise ValueError("per-expert output biases are scaled by the routing weight; the packed format has a single "
                             "unscaled o_bias, so non-zero fc2 biases cannot be converted")
    if f"{prefix}gate.weight" in sd:
        out["w_gate"] = sd[f"{prefix}gate.weight"].clone()
    return out


def unpack_pretrain_experts(keys: torch.Tensor, values: torch.Tensor, bias: Optional[torch.Tensor] = None,
                            w_gate: Optional[torch.Tensor] = None, sub_names: Tuple[str, str] = ("fc1", "fc2"),
                            prefix: str = "", with_zero_out_bias: bool = False) -> Dict[str, torch.Tensor]:
    """Packed pretrain tensors -> LLaVA-format keys (`sub_names` = ("fc1","fc2") for SiglipMLP/CLIPMLP, ("0","2") for the
    Sequential experts).  `with_zero_out_bias` adds zero fc2 biases for modules built with bias=True."""
    E, D, F_ = keys.shape
    if values.shape[0] != E or values.shape[1] != F_:
        raise ValueError(f"keys {tuple(keys.shape)} and values {tuple(values.shape)} do not describe the same experts")
    s1, s2 = sub_names
    sd: Dict[str, torch.Tensor] = {}
    for e in range(E):
        sd[f"{prefix}experts.{e}.{s1}.weight"] = keys[e].t().contiguous()
        if bias is not None:
            sd[f"{prefix}experts.{e}.{s1}.bias"] = bias[e].clone()
        sd[f"{prefix}experts.{e}.{s2}.weight"] = values[e].t().contiguous()
        if with_zero_out_bias:
            sd[f"{prefix}experts.{e}.{s2}.bias"] = torch.zeros(values.shape[2], dtype=values.dtype)
    if w_gate is not None:
        sd[f"{prefix}gate.weight"] = w_gate.clone()
    return sd


def upcycle_from_dense(moelayer, dense_state_dict: Mapping[str, torch.Tensor]) -> None:
    """Sparse upcycling: every expert starts as a copy of one dense MLP (llava_arch.py:110-119)."""
    for expert in moelayer.experts:
        expert.load_state_dict(dense_state_dict)


def remap_dense_to_expert_keys(current: Mapping[str, torch.Tensor], dense: Mapping[str, torch.Tensor],
                               moe_attr: str = "moelayer", dense_attr: str = "mlp") -> Dict[str, torch.Tensor]:
    """The key surgery of llava_arch.py:131-143: for every key `<p>.<moe_attr>.experts.<i>.<sub>.<param>` of the MoE model's
    state dict take the tensor `<p>.<dense_attr>.<sub>.<param>` of the dense checkpoint; other keys keep their value."""
    out = dict(current)
    pat = re.compile(rf"^(?P<p>.*)\.{re.escape(moe_attr)}\.experts\.\d+\.(?P<rest>[^.]+\.(?:weight|bias))$")
    for k in current:
        m = pat.match(k)
        if m:
            src = f"{m.group('p')}.{dense_attr}.{m.group('rest')}"
            if src not in dense:
                raise KeyError(f"dense checkpoint has no '{src}' (needed for '{k}')")
            out[k] = dense[src]
    return out
