"""Checkpoint bridge: per-expert LLaVA modules <-> packed pretrain tensors, checked through the oracle:
the same tokens routed the same way must give the same outputs in both formats.  CPU only."""
import pytest
import torch
import torch.nn as nn

from competesmoe_amd import checkpoint as ck
from oracle import moe_oracle as O


def _experts(E, D, F, bias2_zero=True, seed=0):
    torch.manual_seed(seed)
    ex = nn.ModuleList([nn.Sequential(nn.Linear(D, F), nn.ReLU(), nn.Linear(F, D)) for _ in range(E)])
    if bias2_zero:
        for m in ex:
            nn.init.zeros_(m[2].bias)
    return ex


def test_pack_matches_oracle_both_formats():
    E, D, F, K, B, N = 6, 32, 48, 2, 2, 37
    ex = _experts(E, D, F)
    gate = nn.Linear(D, E, bias=False)
    sd = {f"experts.{k}": v for k, v in ex.state_dict().items()}
    sd["gate.weight"] = gate.weight.detach()
    packed = ck.pack_llava_experts(sd)
    assert packed["keys"].shape == (E, D, F) and packed["values"].shape == (E, F, D) and packed["bias"].shape == (E, F)
    assert torch.equal(packed["w_gate"], gate.weight)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, N, D, generator=g)
    w, idx, _ = O.router_topk(O.gate_logits(x, gate.weight.detach()), K, x.dtype)
    tuples = [(m[0].weight.detach(), m[0].bias.detach(), m[2].weight.detach(), m[2].bias.detach()) for m in ex]
    a = O.compute_moe(x, idx, w, tuples, "relu", D)
    b = O.pretrain_ffn(x, idx.int(), w, packed["keys"], packed["values"], "relu", torch.float32, bias=packed["bias"])
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_round_trip_and_names():
    E, D, F = 4, 16, 24
    ex = _experts(E, D, F, seed=1)
    sd = {f"layer.moelayer.experts.{k}": v for k, v in ex.state_dict().items()}
    packed = ck.pack_llava_experts(sd, prefix="layer.moelayer.")
    back = ck.unpack_pretrain_experts(packed["keys"], packed["values"], packed["bias"], sub_names=("0", "2"),
                                      prefix="layer.moelayer.", with_zero_out_bias=True)
    assert set(back) == set(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k]), k
    ex2 = _experts(E, D, F, seed=9)
    ex2.load_state_dict({k[len("layer.moelayer.experts."):]: v for k, v in back.items()})
    for p, q in zip(ex.parameters(), ex2.parameters()):
        assert torch.equal(p, q)


def test_nonzero_output_bias_is_refused():
    ex = _experts(3, 8, 16, bias2_zero=False)
    with pytest.raises(ValueError, match="output biases"):
        ck.pack_llava_experts({f"experts.{k}": v for k, v in ex.state_dict().items()})
    with pytest.raises(KeyError):
        ck.pack_llava_experts({"gate.weight": torch.zeros(2, 2)})


def test_upcycling_and_key_surgery():
    D, F, E = 8, 12, 3
    dense = nn.Sequential(nn.Linear(D, F), nn.GELU(), nn.Linear(F, D))
    layer = type("L", (), {})()
    layer.experts = _experts(E, D, F, seed=5)
    ck.upcycle_from_dense(layer, dense.state_dict())
    for m in layer.experts:
        for p, q in zip(m.parameters(), dense.parameters()):
            assert torch.equal(p, q)
    cur = {"vision_tower.encoder.layers.3.moelayer.experts.1.fc1.weight": torch.zeros(F, D),
           "vision_tower.encoder.layers.3.moelayer.gate.weight": torch.ones(E, D),
           "vision_tower.encoder.layers.3.layer_norm2.weight": torch.ones(D)}
    dn = {"vision_tower.encoder.layers.3.mlp.fc1.weight": torch.full((F, D), 7.0)}
    out = ck.remap_dense_to_expert_keys(cur, dn)
    assert torch.equal(out["vision_tower.encoder.layers.3.moelayer.experts.1.fc1.weight"], dn["vision_tower.encoder.layers.3.mlp.fc1.weight"])
    assert torch.equal(out["vision_tower.encoder.layers.3.moelayer.gate.weight"], torch.ones(E, D))
    with pytest.raises(KeyError):
        ck.remap_dense_to_expert_keys({"a.moelayer.experts.0.fc2.bias": torch.zeros(1)}, dn)


@pytest.mark.gpu
def test_reference_pretrain_checkpoint_runs_in_llava_layer():
    """A pretrain-format checkpoint (the reference's own golden) unpacked into per-expert modules: the LLaVA `smoe` layer on the
    GPU must reproduce the reference's pretrain output on the same tokens (fp32, 2e-5; the single o_bias is added outside)."""
    import types
    from competesmoe_amd.moe import get_moe
    from tests.golden_util import load, rel_l2
    fx = load("pretrain_smoe_bias_fp32")
    m, st = fx["meta"], fx["state"]
    sd = ck.unpack_pretrain_experts(st["keys"], st["values"], st["bias"], w_gate=st["w_gate"], sub_names=("0", "2"),
                                    with_zero_out_bias=True)
    experts = nn.ModuleList([nn.Sequential(nn.Linear(m["D"], m["F"]), nn.ReLU(), nn.Linear(m["F"], m["D"])) for _ in range(m["E"])])
    args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001)
    layer = get_moe("smoe")(m["D"], m["D"], m["E"], m["K"], experts, args)
    missing, unexpected = layer.load_state_dict(sd, strict=True)
    layer = layer.cuda()
    out = layer(fx["x"].cuda())[0].cpu() + st["o_bias"]
    assert rel_l2(out, fx["output"]) <= 2e-5
    back = ck.pack_llava_experts({k: v.cpu() for k, v in layer.state_dict().items()})
    for k in ("keys", "values", "bias", "w_gate"):
        assert torch.equal(back[k], st[k]), k
