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


def oracle_block(fx, x, lnw, lnb, wg, experts):
    m = fx["meta"]
    args = types.SimpleNamespace(**m["args"])
    with torch.no_grad():
        xn0 = torch.nn.functional.layer_norm(x.detach(), (m["D"],), lnw.detach(), lnb.detach(), m["eps"])
        lg = O.gate_logits(xn0, wg.detach())
        _, gidx, _ = O.router_topk(lg, m["K"], x.dtype)

    def fwd(xn):      # evaluated with the reference's own (tie-broken) indices, as tests/test_oracle_golden.py does
        if m["moe_name"] == "smoe":
            return O.llava_smoe_forward(xn, wg, experts, "gelu_tanh", m["K"], args, forced_idx=fx["selected_experts"])
        return O.llava_competesmoe_forward(xn, wg, experts, "gelu_tanh", m["K"], args, competing=m["competition"],
                                           forced_idx=fx["selected_experts"], forced_aff_idx=fx.get("aff_selected"))

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
    assert abs(float(aux) - float(fx["aux_loss"])) <= 2 * r * max(1.0, abs(float(fx["aux_loss"])))
    ((out.float() * fx["dy"].float()).sum() + aux.float()).backward()
    assert rel_l2(x.grad, fx["x_mid_grad"]) <= 4 * r
    assert rel_l2(lnw.grad, fx["ln_grads"]["weight"]) <= 8 * r
    assert rel_l2(lnb.grad, fx["ln_grads"]["bias"]) <= 8 * r
    assert rel_l2(wg.grad, fx["moe_grads"]["gate.weight"]) <= 8 * r + 1e-4
