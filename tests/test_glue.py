"""Trainer glue: schedule threading across layers, step injection and regulariser aggregation, pinned by the
schedule goldens captured from the reference (tests/golden/*_schedule.pt).  CPU only (no kernels run)."""
import types

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from competesmoe_amd import glue
from tests.golden_util import load


def test_llava_schedule_threading_matches_reference():
    from competesmoe_amd.moe import get_moe
    fx = load("llava_schedule")
    m = fx["meta"]
    args = types.SimpleNamespace(**m["args"])
    mk = lambda: get_moe("competesmoe")(16, 16, 4, 2, nn.Sequential(nn.Linear(16, 8), nn.GELU(), nn.Linear(8, 16)), args)
    n = len(fx["prob_flips"])
    tower = nn.Sequential(*[mk() for _ in range(n - 1)], nn.Identity())
    proj = nn.Sequential(mk())
    plain = get_moe("smoe")(16, 16, 4, 2, nn.Sequential(nn.Linear(16, 8), nn.GELU(), nn.Linear(8, 16)),
                            types.SimpleNamespace(moe_name="smoe", balance_loss_coef=0.01, router_z_loss_coef=0.001))
    tower.add_module("plain", plain)
    torch.manual_seed(m["seed"])                                       # the goldens seed AFTER the layers are built
    out = glue.llava_on_train_begin([tower, proj], m["total_steps"])
    assert sorted(out) == list(range(n))                               # ids counted across both sub-models, `smoe` skipped
    for k, ref in fx["prob_flips"].items():
        assert torch.equal(out[int(k)].cpu(), ref), k
    glue.llava_on_step_end([tower, proj], 7)
    assert all(l.current_steps == 7 for l in list(tower)[: n - 1] + [proj[0]])


def test_pretrain_schedule_threading_and_steps():
    from competesmoe_amd.pretrain import get_moe
    fx = load("pretrain_schedule")
    m = fx["meta"]
    args = types.SimpleNamespace(**m["args"])
    layers = [nn.Sequential(get_moe("competesmoe")(16, 4, 8, n_heads=2, activation=F.relu, log_interval=None, args=args)) for _ in range(len(fx["prob_flips"]))]
    torch.manual_seed(m["seed"])
    prev = glue.pretrain_init_schedules(layers)
    for k, ref in fx["prob_flips"].items():
        assert torch.equal(prev[int(k)].cpu(), ref), k
    model = nn.Sequential(*layers)
    glue.pretrain_set_step(model, 5)
    assert all(l[0].current_steps == 5 for l in layers)


def test_layer_regularizer_sums_scales_and_decays():
    from competesmoe_amd.pretrain.framework_layers import RegularizedLayer

    class L_(RegularizedLayer, nn.Module):
        def __init__(self):
            nn.Module.__init__(self)
            RegularizedLayer.__init__(self)

    a, b = L_(), L_()
    model = nn.Sequential(a, nn.Identity(), b)
    reg = glue.LayerRegularizer(model, stop_after=10, scales={"x": 2.0}, lin_decay={"y"})
    assert a.regularization_present and b.regularization_present
    a.add_reg(lambda: torch.tensor(1.0), "x"); a.add_reg(lambda: torch.tensor(3.0), "x")     # averaged per layer -> 2.0
    b.add_reg(lambda: torch.tensor(4.0), "x"); b.add_reg(lambda: torch.tensor(10.0), "y")
    total, log = reg.get(5)
    assert float(log["x"]) == 6.0 and float(log["y"]) == 10.0                                 # summed over layers, unscaled
    assert float(total) == pytest.approx(6.0 * 2.0 + 10.0 * 0.5)
    t2, log2 = glue.total_loss(torch.tensor(1.5), reg, 0, reg_scale=0.1)
    assert float(t2) == 1.5 and log2 == {}                                                     # accumulators were reset by get()
    with pytest.raises(ValueError):
        glue.LayerRegularizer(model, lin_decay={"y"})
