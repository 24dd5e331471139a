"""CPU: the oracle's restatement of the block around the layer (LayerNorm -> MoE -> residual) against the golden
vectors captured from the reference's SiglipEncoderMoELayer (tests/golden/make_golden_block.py)."""
import types

import pytest
import torch

from oracle import moe_oracle as O
from tests.golden_util import load, rel_l2

CASES = ["smoe", "competesmoe_router", "competesmoe_comp"]


def experts_of(fx, requires_grad=True):
    st, E = fx["moe_state"], fx["meta"]["E"]
    return [tuple(st[f"experts.{i}.{k}"].clone().requires_grad_(requires_grad) for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"))
            for i in range(E)]


def oracle_block(fx, x, lnw, lnb, wg, experts, gate_idx=None, aff_idx=None):
    """`gate_idx` / `aff_idx`: evaluate with these top-k results instead of the reference's recorded ones (the kernel's, where the
    reference's bf16 scores tie exactly: tests/test_block_gpu.py)."""
    m = fx["meta"]
    gate_idx = fx["selected_experts"] if gate_idx is None else gate_idx
    aff_idx = fx.get("aff_selected") if aff_idx is None else aff_idx
    args = types.SimpleNamespace(**m["args"])
    with torch.no_grad():
        xn0 = torch.nn.functional.layer_norm(x.detach(), (m["D"],), lnw.detach(), lnb.detach(), m["eps"])
        lg = O.gate_logits(xn0, wg.detach())
        _, gidx, _ = O.router_topk(lg, m["K"], x.dtype)

    def fwd(xn):      # evaluated with the reference's own (tie-broken) indices, as tests/test_oracle_golden.py does
        if m["moe_name"] == "smoe":
            return O.llava_smoe_forward(xn, wg, experts, "gelu_tanh", m["K"], args, forced_idx=gate_idx)
        return O.llava_competesmoe_forward(xn, wg, experts, "gelu_tanh", m["K"], args, competing=m["competition"],
                                           forced_idx=gate_idx, forced_aff_idx=aff_idx)

    return O.llava_block_forward(x, lnw, lnb, m["eps"], fwd)


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_oracle_block_matches_reference(case, tag):
    fx = load(f"block_{case}_{tag}")
    r = 1e-5 if tag == "fp32" else 2e-3
    x = fx["x_mid"].clone().requires_grad_(True)
    lnw = fx["ln_state"]["weight"].clone().requires_grad_(True)
    lnb = fx["ln_state"]["bias"].clone().requires_grad_(True)
    wg = fx["moe_state"]["gate.weight"].clone().requires_grad_(True)
    experts = experts_of(fx)
    out, aux, infor, st, xn = oracle_block(fx, x, lnw, lnb, wg, experts)
    assert torch.equal(xn.detach(), fx["xn"])
    assert rel_l2(out, fx["output"]) <= r
    assert abs(float(aux.detach()) - float(fx["aux_loss"])) <= 2 * r * max(1.0, abs(float(fx["aux_loss"])))
    ((out.float() * fx["dy"].float()).sum() + aux.float()).backward()
    # bf16, observed: smoe / router step reproduce the reference's bits in the output and in EVERY gradient; the competition step's
    # output and gate gradient too, dx 2.8e-4, LayerNorm 5.1e-4 / 4.2e-4, experts 1.5e-4 (the E dense streams' association)
    g = 4e-5 if tag == "fp32" else (1e-3 if case == "competesmoe_comp" else 1e-6)
    assert rel_l2(x.grad, fx["x_mid_grad"]) <= g
    assert rel_l2(lnw.grad, fx["ln_grads"]["weight"]) <= g
    assert rel_l2(lnb.grad, fx["ln_grads"]["bias"]) <= g
    assert rel_l2(wg.grad, fx["moe_grads"]["gate.weight"]) <= g
    for i, ts in enumerate(experts):
        for t, k in zip(ts, ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")):
            assert rel_l2(t.grad, fx["moe_grads"][f"experts.{i}.{k}"]) <= g, (i, k)


# ------------------------------------------------------------------------------------------------ pretrain stack
@pytest.mark.parametrize("tag", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["smoe", "competesmoe_router"])
def test_oracle_pretrain_block_matches_reference(case, tag):
    """Oracle restatement of the pretrain block against goldens captured from the reference's RelativeMoeTransformerEncoderLayer
    (tests/golden/make_golden_pretrain_block.py): fp32, and bf16 autocast on an fp32 residual stream."""
    fx = load(f"pretrain_block_{case}_{tag}")
    m, st = fx["meta"], fx["state"]
    op = torch.bfloat16 if m["bf16"] else torch.float32
    r = 1e-5 if tag == "fp32" else 4e-3
    x = fx["mid"].clone().requires_grad_(True)
    lnw = fx["norm2"]["weight"].clone().requires_grad_(True)
    lnb = fx["norm2"]["bias"].clone().requires_grad_(True)
    ps = {k: st[k].clone().requires_grad_(True) for k in ("w_gate", "keys", "values")}
    reg = {}

    def moe(xn):
        xx = xn.to(op)
        lg = O.gate_logits(xx, ps["w_gate"].to(op))
        w, idx, _ = O.router_topk(lg, m["K"], xn.dtype)
        reg["mlp_ebalance"] = O.entropy_balance(lg) * m["args"]["balance_loss_coef"]
        return O.pretrain_ffn(xn, idx, w, ps["keys"], ps["values"], "relu", op)

    out, moe_out, xn = O.pretrain_block_forward(x, lnw, lnb, fx["eps"], moe)
    assert out.dtype == torch.float32 and moe_out.dtype == op and xn.dtype == torch.float32
    assert rel_l2(out, fx["output"]) <= r, rel_l2(out, fx["output"])
    assert abs(float(reg["mlp_ebalance"]) - float(fx["reg_loss"]["mlp_ebalance"])) <= 1e-5
    ((out.float() * fx["dy"]).sum() + reg["mlp_ebalance"].float()).backward()
    assert rel_l2(x.grad, fx["mid_grad"]) <= 4 * r
    assert rel_l2(lnw.grad, fx["norm2_grads"]["weight"]) <= 8 * r
    assert rel_l2(lnb.grad, fx["norm2_grads"]["bias"]) <= 8 * r
    for k, p in ps.items():
        assert rel_l2(p.grad, fx["grads"][k]) <= 8 * r + (1e-4 if k == "w_gate" else 0), k


def oracle_pretrain_block_deepseek(fx, idx):
    """The oracle's pretrain block around a `deepseekv2/3` layer evaluated with the given top-k indices (the reference's own, recorded
    in the fixture, to PIN it; the kernel's, in tests/test_pretrain_block_gpu.py, where the reference's bf16 scores tie exactly).
    Returns (out, reg, x_grad, {"norm2.weight" / "norm2.bias" / parameter name: grad})."""
    m, st = fx["meta"], fx["state"]
    op = torch.bfloat16 if m["bf16"] else torch.float32
    x = fx["mid"].clone().requires_grad_(True)
    lnw = fx["norm2"]["weight"].clone().requires_grad_(True)
    lnb = fx["norm2"]["bias"].clone().requires_grad_(True)
    ps = {k: v.clone().requires_grad_(True) for k, v in st.items() if v.is_floating_point()}
    reg = {}

    def moe(xn):
        out, lg = O.pretrain_deepseek_forward(xn, ps["w_gate"], ps["keys"], ps["values"], ps["keys_shared"], ps["values_shared"],
                                              m["K"], m["moe_name"], op, xn.dtype, forced_idx=idx.long().view(m["B"], m["N"], -1))
        reg["mlp_ebalance"] = O.entropy_balance(lg) * m["args"]["balance_loss_coef"]
        return out

    out, _, _ = O.pretrain_block_forward(x, lnw, lnb, fx["eps"], moe)
    ((out.float() * fx["dy"]).sum() + reg["mlp_ebalance"].float()).backward()
    grads = {k: p.grad for k, p in ps.items()}
    grads["norm2.weight"], grads["norm2.bias"] = lnw.grad, lnb.grad
    return out.detach(), reg, x.grad, grads


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_oracle_pretrain_block_deepseekv3_is_pinned_by_the_reference(tag):
    """deepseekv3 inside the pretrain block (shared expert + sigmoid top-k), with the indices the reference's own torch.topk returned:
    the oracle reproduces the fixture, gradients included -- which makes it a legitimate reference for the rows where the kernel
    breaks an exact tie of bf16 sigmoids the other way."""
    fx = load(f"pretrain_block_deepseekv3_{tag}")
    out, reg, xg, grads = oracle_pretrain_block_deepseek(fx, fx["selected_experts"])
    r = 1e-5 if tag == "fp32" else 1e-6
    assert rel_l2(out, fx["output"]) <= r, rel_l2(out, fx["output"])
    assert abs(float(reg["mlp_ebalance"]) - float(fx["reg_loss"]["mlp_ebalance"])) <= 1e-5
    errs = {"dx": rel_l2(xg, fx["mid_grad"]), "norm2.weight": rel_l2(grads["norm2.weight"], fx["norm2_grads"]["weight"]),
            "norm2.bias": rel_l2(grads["norm2.bias"], fx["norm2_grads"]["bias"])}
    for k, g in fx["grads"].items():
        if g is not None:
            errs[k] = rel_l2(grads[k], g)
    print(tag, {k: f"{v:.2e}" for k, v in errs.items()})
    # observed: fp32 <= 2.6e-7; bf16 dx and d w_gate the reference's bits, everything else <= 1.2e-7
    assert all(v <= (4e-5 if tag == "fp32" else 1e-5) for v in errs.values()), errs
