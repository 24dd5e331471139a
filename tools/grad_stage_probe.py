#!/usr/bin/env python3
"""Where the pretrain smoe layer's bf16 gradients leave the reference's: the CPU oracle (pinned by the reference's goldens) and the
HIP path side by side on a golden's inputs, gradients compared stage by stage (d weights, d logits, d w_gate, d x).
usage (GPU box): python tools/grad_stage_probe.py [fixture, default pretrain_smoe_bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_util import load, rel_l2  # noqa: E402
from oracle import moe_oracle as O  # noqa: E402
from competesmoe_amd import _lib as L  # noqa: E402
from competesmoe_amd.functional import GateSelect, MoEFFNPacked  # noqa: E402


def main():
    fx = load(sys.argv[1] if len(sys.argv) > 1 else "pretrain_smoe_bf16")
    meta, st = fx["meta"], fx["state"]
    K, coef = meta["K"], meta["args"]["balance_loss_coef"]
    op = torch.bfloat16
    # ---- oracle
    x = fx["x"].clone().requires_grad_(True)
    wg = st["w_gate"].clone().requires_grad_(True)
    keys, values = st["keys"].clone().requires_grad_(True), st["values"].clone().requires_grad_(True)
    lg = O.gate_logits(x.to(op), wg.to(op))
    lg.retain_grad()
    w, idx, sm = O.router_topk(lg, K, x.dtype)
    w.retain_grad()
    out = O.pretrain_ffn(x, idx, w, keys, values, "relu", op)
    reg = O.entropy_balance(lg) * coef
    ((out.float() * fx["dy"]).sum() + reg.float()).backward()
    # ---- HIP
    dev = "cuda"
    xg = fx["x"].to(dev).requires_grad_(True)
    wgg = st["w_gate"].to(dev).requires_grad_(True)
    kg, vg = st["keys"].to(dev).requires_grad_(True), st["values"].to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=op):
        x2 = xg.reshape(-1, xg.shape[-1]).to(op)
        lg2, sm2, idx2, w2 = GateSelect.apply(x2, wgg, K, L.SEL_SOFTMAX, False)
        lg2.retain_grad()
        w2.retain_grad()
        wk = w2.float().contiguous()
        out2 = MoEFFNPacked.apply(x2, wk, idx2, kg, vg, None, None, L.ACT_RELU, L.COMBINE_DOT, None, None)
        reg2 = O.entropy_balance(lg2.view(lg.shape)) * coef
    ((out2.float().view(fx["dy"].shape) * fx["dy"].to(dev)).sum() + reg2.float()).backward()
    c = lambda t: t.detach().float().cpu()
    print("routing equal:", bool((c(idx2).view(idx.shape) == idx).all()))
    print("out      ", rel_l2(c(out2).view(out.shape), out.detach().float()))
    print("d weights", rel_l2(c(w2.grad).view(w.shape), w.grad.float()))
    print("d logits ", rel_l2(c(lg2.grad).view(lg.shape), lg.grad.float()))
    print("d w_gate ", rel_l2(c(wgg.grad), wg.grad.float()))
    print("d keys   ", rel_l2(c(kg.grad), keys.grad.float()), " d values", rel_l2(c(vg.grad), values.grad.float()))
    print("d x      ", rel_l2(c(xg.grad), x.grad.float()))


if __name__ == "__main__":
    main()
