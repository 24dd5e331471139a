#!/usr/bin/env python3
"""Observed parity of the pretrain block (src + pkm(norm2(src)), fp32 stream under bf16 autocast) against its goldens, quantity by
quantity -- the numbers behind the tolerances of tests/test_pretrain_block_gpu.py.   usage (GPU box): python tools/block_parity_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_util import load, rel_l2  # noqa: E402
import tests.test_pretrain_block_gpu as TB  # noqa: E402

DEV = "cuda"


def llava_blocks():
    import tests.test_block_gpu as TL
    from competesmoe_amd import ops
    for case in ("smoe", "competesmoe_router", "competesmoe_comp"):
        fx = load(f"block_{case}_bf16")
        m = fx["meta"]
        blk = TL.build_block(fx, torch.bfloat16)
        x = fx["x_mid"].to(DEV).requires_grad_(True)
        out, aux, _ids, infor = blk(x)
        ((out.float() * fx["dy"].to(DEV).float()).sum() + aux.float()).backward()
        with torch.no_grad():
            xn, _, _, lg = ops.layernorm_gate(fx["x_mid"].to(DEV).reshape(-1, m["D"]), blk.layer_norm2.weight, blk.layer_norm2.bias,
                                              m["eps"], blk.moelayer.gate.weight)
        o, go = out.detach().cpu().reshape(-1, m["D"]).double(), fx["output"].reshape(-1, m["D"]).double()
        row_err = (o - go).norm(dim=-1) / (go.norm(dim=-1) + 1e-12)
        bad = row_err > 5e-2
        line = [f"LLaVA block {case}: xn {rel_l2(xn.cpu(), fx['xn'].reshape(-1, m['D'])):.2e} logits {rel_l2(lg.cpu(), fx['gate_logits'].reshape(-1, m['E'])):.2e}",
                f"rows routed differently {float(bad.float().mean()):.4f}, others {rel_l2(o[~bad], go[~bad]):.2e}",
                f"aux {float(aux):.6f}/{float(fx['aux_loss']):.6f}", f"d x {rel_l2(x.grad.cpu(), fx['x_mid_grad']):.2e}",
                f"d ln.w {rel_l2(blk.layer_norm2.weight.grad.cpu(), fx['ln_grads']['weight']):.2e} d ln.b {rel_l2(blk.layer_norm2.bias.grad.cpu(), fx['ln_grads']['bias']):.2e}",
                f"d gate {rel_l2(blk.moelayer.gate.weight.grad.cpu(), fx['moe_grads']['gate.weight']):.2e}"]
        worst = 0.0
        for k, gref in fx["moe_grads"].items():
            if gref is None or not k.startswith("experts."):
                continue
            pp = dict(blk.moelayer.named_parameters())[k]
            worst = max(worst, rel_l2(pp.grad.cpu(), gref))
        line.append(f"worst expert grad {worst:.2e}")
        print(" | ".join(line))


def main():
    llava_blocks()
    for case in TB.CASES:
        fx = load(f"pretrain_block_{case}_bf16")
        blk, layer, kw = TB.build(fx)
        x = fx["mid"].to(DEV).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = blk(x, **kw)
            reg = layer.get_reg_loss()
        gold = fx["output"].to(DEV)
        o2 = (out.detach() - x.detach()).reshape(-1, out.shape[-1]).double()
        g2 = (gold - x.detach()).reshape(-1, out.shape[-1]).double()
        row_err = (o2 - g2).norm(dim=-1) / (g2.norm(dim=-1) + 1e-12)
        bad = row_err > 5e-2
        line = [f"{case}: rows routed differently {float(bad.float().mean()):.4f}, others {rel_l2(o2[~bad], g2[~bad]):.2e}"]
        ((out.float() * fx["dy"].to(DEV)).sum() + sum(v.float() for v in reg.values())).backward()
        line.append(f"d mid {rel_l2(x.grad, fx['mid_grad'].to(DEV)):.2e}")
        line.append(f"d ln.w {rel_l2(blk.norm2.weight.grad, fx['norm2_grads']['weight'].to(DEV)):.2e} d ln.b {rel_l2(blk.norm2.bias.grad, fx['norm2_grads']['bias'].to(DEV)):.2e}")
        for name in ("keys", "values", "w_gate"):
            g = fx["grads"].get(name)
            p = getattr(layer, name, None)
            if g is not None and p is not None and p.grad is not None:
                line.append(f"d {name} {rel_l2(p.grad, g.to(DEV)):.2e}")
        print(" | ".join(line))


if __name__ == "__main__":
    main()
