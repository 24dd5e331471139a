"""CPU: a freshly constructed layer carries the reference's initial state.

`init_gate_weights` (moe_model/model/moe/moe.py:50-70) draws `gate.weight ~ N(0, 0.02)` from a generator seeded with 42, so
every layer of every run starts from the same gate.  The LLaVA goldens were captured from reference layers that went through that
initialiser; nothing here loads the golden's state dict -- the layer under test is built from the constructor alone and must
hold the same numbers.  Constructing a layer needs the C-ABI library to load but launches nothing."""
import types

import pytest
import torch
import torch.nn as nn

from tests.golden_util import load, args_of


def _experts(E, D, F, Dout):
    return nn.ModuleList([nn.Sequential(nn.Linear(D, F), nn.GELU(), nn.Linear(F, Dout)) for _ in range(E)])


@pytest.mark.parametrize("case", ["smoe", "competesmoe_router", "competesmoe_comp", "smoe_share", "deepseekv3", "smoe_proj"])
def test_fresh_llava_layer_has_the_reference_seed42_gate(case):
    from competesmoe_amd.moe import get_moe
    fx = load(f"llava_{case}_fp32")
    m, args = fx["meta"], args_of(fx)
    D, F, Dout, E, K = m["D"], m["F"], m["Dout"], m["E"], m["K"]
    torch.manual_seed(987654)            # the global RNG state must not matter: the gate has its own generator
    if m["moe_name"] in ("smoe_share", "deepseekv3"):
        layer = get_moe(m["moe_name"])(D, Dout, E, K, _experts(1, D, F, Dout)[0], args)
    else:
        layer = get_moe(m["moe_name"])(D, Dout, E, K, _experts(E, D, F, Dout), args)
    gold = fx["state"]["gate.weight"]
    assert layer.gate.weight.shape == gold.shape
    assert torch.equal(layer.gate.weight.detach(), gold), "fresh gate differs from the reference's seed-42 draw"
    # and the bf16 copy of the same layer is the rounded draw (the bf16 goldens were cast after construction)
    gold16 = load(f"llava_{case}_bf16")["state"]["gate.weight"]
    assert torch.equal(layer.gate.weight.detach().to(torch.bfloat16), gold16)


def test_init_weight_false_leaves_the_default_linear_init():
    from competesmoe_amd.moe import get_moe
    fx = load("llava_smoe_fp32")
    m = fx["meta"]
    a = dict(m["args"])
    a["init_weight"] = False
    torch.manual_seed(3)
    layer = get_moe("smoe")(m["D"], m["Dout"], m["E"], m["K"], _experts(m["E"], m["D"], m["F"], m["Dout"]), types.SimpleNamespace(**a))
    assert not torch.equal(layer.gate.weight.detach(), fx["state"]["gate.weight"])


def test_fresh_pretrain_layer_reproduces_the_reference_initialisers():
    """moe_pretrain_model/layers/moe/moe.py:120-127: w_gate, keys ~ N(0, D^-0.5), values ~ N(0, (E*F)^-0.5), drawn from the global
    RNG in the order w_gate, keys, values.  The golden's generator seeds the global RNG with `seed` (0) right before construction."""
    import torch.nn.functional as F
    from competesmoe_amd.pretrain import get_moe
    fx = load("pretrain_smoe_fp32")
    m = fx["meta"]
    torch.manual_seed(0)
    layer = get_moe("smoe")(m["D"], m["E"], m["F"], n_heads=m["K"], activation=F.relu, log_interval=None,
                            args=types.SimpleNamespace(**m["args"]))
    for k in ("w_gate", "keys", "values"):
        assert torch.equal(getattr(layer, k).detach(), fx["state"][k]), k


def test_same_length_schedule_load_refreshes_the_branch_test():
    """resume order is set_total_steps() then load_state_dict() copying a checkpointed `prob_flips` of the SAME length in
    place; the layer must follow the checkpointed schedule (the reference reads prob_flips[i].item() every step)."""
    from competesmoe_amd.moe import get_moe
    fx = load("llava_competesmoe_comp_fp32")
    m, args = fx["meta"], args_of(fx)
    layer = get_moe("competesmoe")(m["D"], m["Dout"], m["E"], m["K"], _experts(m["E"], m["D"], m["F"], m["Dout"]), args)
    torch.manual_seed(5)
    layer.set_total_steps(10, 0, {})
    layer.set_current_steps(layer.step_warm + 2)
    x = torch.zeros(1, 2, m["D"], requires_grad=True)
    before = layer._competing(x)
    flipped = layer.prob_flips.clone()
    flipped[2] = not bool(flipped[2])
    sd = layer.state_dict()
    sd["prob_flips"] = flipped
    layer.load_state_dict(sd)                       # same length: copied in place, the host copy must notice
    assert layer._competing(x) == (not before)
    # pretrain flavour: the schedule lives in prob_flips_final[id_layer]
    import torch.nn.functional as F
    from competesmoe_amd.pretrain import get_moe as get_pre
    pf = load("pretrain_competesmoe_comp_fp32")
    pm = pf["meta"]
    pl = get_pre("competesmoe")(pm["D"], pm["E"], pm["F"], n_heads=pm["K"], activation=F.relu, log_interval=None,
                                args=types.SimpleNamespace(**pm["args"]))
    pl.set_total_steps(id_layer=0)
    pl.set_current_steps(pl.step_warm + 1)
    b0 = pl._competing(x, 0)
    pl.prob_flips_final[0][1] = not bool(pl.prob_flips_final[0][1])      # in-place edit, same length
    assert pl._competing(x, 0) == (not b0)
