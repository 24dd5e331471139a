#!/usr/bin/env python3
"""Observed parity of the HIP path against every layer-level golden fixture, as one table (GPU box).

Prints, per fixture: relative L2 of the output, the fraction of rows routed differently from the reference (rows whose error is
O(1)), relative L2 over the rows routed alike, the aux / regularisation losses, and relative L2 of every gradient.  The numbers
calibrate the tolerances written in tests/test_*_gpu.py and are kept under profiles/rNN/parity_report.txt.

    python tools/parity_report.py [--softplus-precise]   (CSMOE_SOFTPLUS_PRECISE=1: the A/B of the affinity kernel's exp / log)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--softplus-precise" in sys.argv:
    os.environ["CSMOE_SOFTPLUS_PRECISE"] = "1"

import torch  # noqa: E402

from tests.golden_util import load, rel_l2  # noqa: E402

DEV = "cuda"


def rows(out, gold):
    o2, g2 = out.detach().reshape(-1, out.shape[-1]).double(), gold.reshape(-1, out.shape[-1]).double().to(out.device)
    err = (o2 - g2).norm(dim=-1) / (g2.norm(dim=-1) + 1e-12)
    bad = err > 5e-2
    same = rel_l2(o2[~bad], g2[~bad]) if (~bad).any() else float("nan")
    return float(bad.float().mean()), same


def llava():
    import tests.test_llava_modules_gpu as TL
    print("== LLaVA stack (moe_model/model/moe) ==")
    for case in TL.CASES + ["competesmoe_comp_normsigmoid"]:
        for tag in ("fp32", "bf16"):
            try:
                fx = load(f"llava_{case}_{tag}")
            except FileNotFoundError:
                continue
            layer, dt = TL.build_layer(fx)
            x = fx["x"].to(DEV).requires_grad_(True)
            out, aux, _, infor = layer(x)
            bad, same = rows(out, fx["output"])
            line = [f"{case}_{tag}: out {rel_l2(out, fx['output'].to(DEV)):.2e} routed-differently {bad:.4f} same-rows {same:.2e}",
                    f"aux {float(aux):.6f}/{float(fx['aux_loss']):.6f}"]
            if "aff_selected" in fx and hasattr(layer, "competition_policy"):
                with torch.no_grad():
                    r = layer.competition_policy(fx["x"].to(DEV))
                idx = r[1].cpu().long().reshape(-1, r[1].shape[-1])
                gi = fx["aff_selected"].reshape(idx.shape)
                ga = fx["aff_scores"].reshape(idx.shape[0], -1).float()
                aff = r[3].cpu().float().reshape(ga.shape)
                mism = (idx.sort(-1).values != gi.sort(-1).values).any(-1)
                K = idx.shape[-1]
                srt = ga.sort(-1, descending=True).values
                tie = srt[:, K - 1] == srt[:, K]                    # the reference's own scores tie at the top-K boundary
                ulp = (aff != ga).any(-1)                          # our affinity differs from the reference's somewhere in the row
                line.append(f"affinity-topk-set-mismatch {float(mism.float().mean()):.4f} "
                            f"(of those: exact tie in the reference's scores {int((mism & tie).sum())}, "
                            f"1-ulp affinity difference {int((mism & ~tie & ulp).sum())}, other {int((mism & ~tie & ~ulp).sum())}; "
                            f"rows with any affinity difference {int(ulp.sum())}/{ulp.numel()})")
            ((out.float() * fx["dy"].to(DEV).float()).sum() + aux.float()).backward()
            line.append(f"dx {rel_l2(x.grad, fx['x_grad'].to(DEV)):.2e}")
            worst = 0.0
            for k, p in layer.named_parameters():
                g = fx["grads"].get(k)
                if g is not None and p.grad is not None:
                    worst = max(worst, rel_l2(p.grad, g.to(DEV)))
            line.append(f"worst-param-grad {worst:.2e}")
            if bad > 0:
                # rows routed unlike the reference's run (exact ties of its bf16 scores): the numbers above compare different routings;
                # the comparison that means something is with the pinned oracle evaluated under the KERNEL's indices
                with torch.no_grad():
                    gidx = layer.topk_expert(layer.gate_logits(fx["x"].to(DEV)))[1].reshape(-1, layer.num_selected).cpu().long()
                    aidx = layer.competition_policy(fx["x"].to(DEV))[1].reshape(gidx.shape[0], -1).cpu().long() \
                        if hasattr(layer, "competition_policy") else None
                oxg, og = TL.oracle_grads(fx, gidx, aidx)
                w2 = max(rel_l2(p.grad.cpu(), og[k]) for k, p in layer.named_parameters() if og.get(k) is not None and p.grad is not None)
                line.append(f"AGAINST THE ORACLE UNDER THE KERNEL'S INDICES: dx {rel_l2(x.grad.cpu(), oxg):.2e} worst-param-grad {w2:.2e}")
            print("  " + " | ".join(line))


def pretrain():
    import tests.test_pretrain_modules_gpu as TP
    print("== pretrain stack (moe_pretrain_model/layers/moe), goldens from the reference's Triton kernels ==")
    names = [f"{c}_{t}" for c in TP.CASES for t in ("fp32", "bf16")] + \
        ["competesmoe_cosine_fp32", "competesmoe_normweight_fp32", "competesmoe_normsigmoid_fp32", "competesmoe_comp_intopk_fp32",
         "competesmoe_comp_tribrid_fp32"]
    for name in names:
        fx = load(f"pretrain_{name}")
        layer, kw = TP.build(fx)
        bf16 = fx["meta"]["bf16"]
        x = fx["x"].to(DEV).requires_grad_(True)
        spy = {}
        if hasattr(layer, "ffn"):
            ffn0 = layer.ffn
            layer.ffn = lambda xx, sel, ww, *a, _f=ffn0, **k: (spy.setdefault("idx", sel.detach().cpu().long()), _f(xx, sel, ww, *a, **k))[1]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            out = layer(x, **kw)
            reg = layer.get_reg_loss()
        bad, same = rows(out, fx["output"])
        line = [f"{name}: out {rel_l2(out, fx['output'].to(DEV)):.2e} routed-differently {bad:.4f} same-rows {same:.2e}"]
        line.append("reg " + " ".join(f"{k}={float(v):.7f}/{float(fx['reg_loss'][k]):.7f}" for k, v in reg.items()))
        if "aff_selected" in fx:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
                r = layer.competition_policy_mlp_faster(fx["x"].to(DEV))
            mism = (r[1].cpu().long().sort(-1).values != fx["aff_selected"].sort(-1).values).any(-1)
            line.append(f"affinity-topk-set-mismatch {float(mism.float().mean()):.4f} aff {rel_l2(r[3].cpu().float(), fx['aff_scores'].float()):.2e}")
        ((out.float() * fx["dy"].to(DEV)).sum() + sum(v.float() for v in reg.values())).backward()
        line.append(f"dx {rel_l2(x.grad, fx['x_grad'].to(DEV)):.2e}")
        line.append("grads " + " ".join(f"{k}={rel_l2(p.grad, fx['grads'][k].to(DEV)):.2e}" for k, p in layer.named_parameters()
                                        if fx["grads"].get(k) is not None and p.grad is not None))
        if bad > 0 and "idx" in spy and not fx["meta"].get("competition"):
            oxg, og = TP.oracle_grads(fx, spy["idx"])
            line.append("AGAINST THE ORACLE UNDER THE KERNEL'S INDICES: dx " + f"{rel_l2(x.grad.cpu(), oxg):.2e} grads " +
                        " ".join(f"{k}={rel_l2(p.grad.cpu(), og[k]):.2e}" for k, p in layer.named_parameters()
                                 if og.get(k) is not None and p.grad is not None))
        print("  " + " | ".join(line))


if __name__ == "__main__":
    print("CSMOE_SOFTPLUS_PRECISE =", os.environ.get("CSMOE_SOFTPLUS_PRECISE", "0"))
    llava()
    pretrain()
