"""GPU: the expert-parallel layer gives the single-GPU layer's numbers.

(1) world_size 1 over RCCL ("nccl") on the one GPU of the box: EPSMoeLayer == SMoeLayer bit for bit.
(2) world_size 2, both ranks on cuda:0 (RCCL refuses two ranks on one device, so for THIS test only the two collectives
    are carried by gloo through host memory): each rank's EP output equals the full single-GPU layer on its tokens, and the
    local expert gradients equal the sum over ranks of the single-GPU gradients."""
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _experts(E, D, F, seed, dt, dev):
    torch.manual_seed(seed)
    ex = nn.ModuleList([nn.Sequential(nn.Linear(D, F), nn.GELU(), nn.Linear(F, D)) for _ in range(E)])
    return ex.to(dev).to(dt)


def _run(rank, world, port, backend, dt_name, q, chunks=1, empty_half=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from competesmoe_amd import ep
        from competesmoe_amd.moe import get_moe
        if backend == "gloo":
            real = dist.all_to_all_single

            def via_host(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
                o = torch.empty(out.shape, dtype=out.dtype)
                real(o, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
                out.copy_(o)
            dist.all_to_all_single = via_host
            real_direct = ep.exchange_direct

            def direct_via_host(dst, src, group=None):       # gloo moves host tensors: stage the (peer, expert) messages through the host
                src_h = [(p_, v.cpu()) for p_, v in src]
                dst_h = [(p_, torch.empty(v.shape, dtype=v.dtype)) for p_, v in dst]
                real_direct(dst_h, src_h, group).wait()
                for (_, d_), (_, h_) in zip(dst, dst_h):
                    if d_.shape[0]:
                        d_.copy_(h_)
                return ep._Works([])
            ep.exchange_direct = direct_via_host
        dt = torch.float32 if dt_name == "fp32" else torch.bfloat16
        B, N, D, F, E, K = 2, 96, 64, 128, 8, 2
        args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001)
        full = get_moe("smoe")(D, D, E, K, _experts(E, D, F, 7, dt, dev), args).to(dev).to(dt).train()
        El = E // world
        local = nn.ModuleList([_experts(E, D, F, 7, dt, dev)[rank * El + i] for i in range(El)])
        epl = ep.EPSMoeLayer(D, D, E, K, local, args, chunks=chunks).to(dev).to(dt).train()
        g = torch.Generator().manual_seed(50 + rank)
        x = torch.randn(B, N, D, generator=g)
        if empty_half:      # experts E/2 .. E-1 are never selected: whole groups (and, at world 2, a whole rank) receive no rows
            x[..., 0] = x[..., 0].abs() + 4.0
            with torch.no_grad():
                for lay in (full, epl):
                    lay.gate.weight[: E // 2, 0] = 1.0
                    lay.gate.weight[E // 2:, 0] = -1.0
        x = x.to(dt).to(dev)
        dy = torch.randn(B, N, D, generator=g).to(dt).to(dev)
        xa = x.clone().requires_grad_(True)
        xb = x.clone().requires_grad_(True)
        oa, aa, _, _ = full(xa)
        torch.autograd.backward([oa, aa.float()], [dy, torch.ones((), device=dev)])
        ob, ab, _, _ = epl(xb)
        torch.autograd.backward([ob, ab.float()], [dy, torch.ones((), device=dev)])
        torch.cuda.synchronize()
        ok = torch.equal(oa, ob) and float(aa) == float(ab)
        ok = ok and torch.equal(xa.grad, xb.grad)
        # expert grads: EP local grads == sum over ranks of the single-GPU grads for those experts
        tol = 1e-5 if dt == torch.float32 else 2e-2
        for i in range(El):
            for (n1, p1), (n2, p2) in zip(full.experts[rank * El + i].named_parameters(), epl.experts[i].named_parameters()):
                pass
        refs = []
        for e in range(E):
            for p in full.experts[e].parameters():
                gsum = p.grad.detach().float().cpu()
                if world > 1:
                    dist.all_reduce(gsum)     # gloo, host tensors
                refs.append(gsum)
        mine = [p.grad.detach().float().cpu() for i in range(El) for p in epl.experts[i].parameters()]
        per = len(list(full.experts[0].parameters()))
        for j, gm in enumerate(mine):
            gr = refs[rank * El * per + j]
            err = float((gm - gr).norm() / (gr.norm() + 1e-12))
            ok = ok and err <= tol
        # replicated gate: EP grad is the all-reduced sum
        gg = full.gate.weight.grad.detach().float().cpu()
        if world > 1:
            dist.all_reduce(gg)
        ok = ok and float((epl.gate.weight.grad.float().cpu() - gg).norm() / gg.norm()) <= tol
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run_comp(rank, world, port, backend, dt_name, q, *unused):
    """competesmoe_ep on a competition step and on a router step against the single-GPU `competesmoe` on the same tokens."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from competesmoe_amd import ep
        from competesmoe_amd.moe import get_moe
        if backend == "gloo":          # two ranks on ONE device: RCCL refuses, the collectives go through host memory for this test
            real_a2a, real_ag = dist.all_to_all_single, dist.all_gather

            def a2a_host(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
                o = torch.empty(out.shape, dtype=out.dtype)
                real_a2a(o, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
                out.copy_(o)

            def ag_host(outs, inp, group=None):
                hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
                real_ag(hs, inp.cpu(), group=group)
                for o, h in zip(outs, hs):
                    o.copy_(h)
            dist.all_to_all_single, dist.all_gather = a2a_host, ag_host
        dt = torch.float32 if dt_name == "fp32" else torch.bfloat16
        B, N, D, F, E, K = 2, 64, 64, 128, 8, 2
        args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001, rate_flip=1.0, warm_up=0.0, max_compete_in_iter=8,
                                     router_loss_coef=0.03, diversity_loss_coef=0.02, bal_comp_loss_coef=0.01, hybrid=True,
                                     router_theta=0.5, moe_name="competesmoe")
        full = get_moe("competesmoe")(D, D, E, K, _experts(E, D, F, 7, dt, dev), args).to(dev).to(dt).train()
        El = E // world
        local = nn.ModuleList([_experts(E, D, F, 7, dt, dev)[rank * El + i] for i in range(El)])
        epl = ep.EPCompeteSMoE(D, D, E, K, local, args).to(dev).to(dt).train()
        ok = True
        for competing in (True, False):
            for lay in (full, epl):
                lay.total_steps, lay.step_warm, lay.flip_steps = 8, 0, 8
                lay.prob_flips = torch.full((8,), competing, dtype=torch.bool, device=dev)
                lay._flips_host = None
                lay.set_current_steps(2)
                for p in lay.parameters():
                    p.grad = None
            g = torch.Generator().manual_seed(60 + rank)
            x = torch.randn(B, N, D, generator=g).to(dt).to(dev)
            dy = torch.randn(B, N, D, generator=g).to(dt).to(dev)
            xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            oa, aa, _, ia = full(xa)
            torch.autograd.backward([oa, aa.float()], [dy, torch.ones((), device=dev)])
            ob, ab, _, ib = epl(xb)
            torch.autograd.backward([ob, ab.float()], [dy, torch.ones((), device=dev)])
            torch.cuda.synchronize()
            tol = 2e-5 if dt == torch.float32 else 2e-2
            rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
            ok = ok and torch.equal(oa, ob) and set(ia) == set(ib)
            ok = ok and abs(float(aa) - float(ab)) <= 1e-5 * max(1.0, abs(float(aa))) * (1 if dt == torch.float32 else 100)
            ok = ok and rel(xb.grad, xa.grad) <= tol
            per = len(list(full.experts[0].parameters()))
            refs = []
            for e in range(E):
                for p in full.experts[e].parameters():
                    gsum = p.grad.detach().float().cpu()
                    if world > 1:
                        dist.all_reduce(gsum)
                    refs.append(gsum)
            mine = [p.grad.detach().float().cpu() for i in range(El) for p in epl.experts[i].parameters()]
            for j, gm in enumerate(mine):
                ok = ok and rel(gm, refs[rank * El * per + j]) <= tol
            gg = full.gate.weight.grad.detach().float().cpu()
            if world > 1:
                dist.all_reduce(gg)
            ok = ok and rel(epl.gate.weight.grad.cpu(), gg) <= tol
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run_pretrain(rank, world, port, backend, dt_name, q, chunks=1, empty_half=False):
    """Pretrain `smoe_ep` against the single-GPU pretrain `smoe` on the same tokens.  dt_name: "fp32", "bf16" (bf16 autocast over fp32
    masters: the stack's training configuration; ReLU experts, so d w comes out of the dH launch's dot epilogue), "bf16_bias"."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import torch.nn.functional as F
        from competesmoe_amd import ep
        from competesmoe_amd.pretrain import get_moe
        if backend == "gloo":
            real = dist.all_to_all_single

            def via_host(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
                o = torch.empty(out.shape, dtype=out.dtype)
                real(o, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
                out.copy_(o)
            dist.all_to_all_single = via_host
            real_direct = ep.exchange_direct

            def direct_via_host(dst, src, group=None):
                src_h = [(p_, v.cpu()) for p_, v in src]
                dst_h = [(p_, torch.empty(v.shape, dtype=v.dtype)) for p_, v in dst]
                real_direct(dst_h, src_h, group).wait()
                for (_, d_), (_, h_) in zip(dst, dst_h):
                    if d_.shape[0]:
                        d_.copy_(h_)
                return ep._Works([])
            ep.exchange_direct = direct_via_host
        bias = dt_name.endswith("_bias")
        block = dt_name.endswith("_block")     # inside pretrain.MoEBlock: fp32 stream, fused LayerNorm + gate, residual in the combine
        autocast = dt_name.startswith("bf16")
        B, N, D, Fh, E, K = 2, 96, 64, 128, 8, 2
        if dt_name.endswith("_big"):        # enough rows per group for the kernels that convert the fp32 masters in their tile fill
            B, N, D, Fh = 4, 1024, 256, 256
        args = types.SimpleNamespace(balance_loss_coef=0.01)
        torch.manual_seed(11)
        full = get_moe("smoe")(D, E, Fh, n_heads=K, activation=F.relu, bias=bias, args=args).to(dev).train()
        if bias:
            with torch.no_grad():
                full.bias.normal_(0, 0.1)
                full.o_bias.normal_(0, 0.1)
        El = E // world
        epl = get_moe("smoe_ep")(D, E, Fh, n_heads=K, activation=F.relu, bias=bias, args=args, chunks=chunks).to(dev).train()
        assert tuple(epl.keys.shape) == (El, D, Fh) and tuple(epl.values.shape) == (El, Fh, D)
        full.regularization_present = epl.regularization_present = True
        with torch.no_grad():
            epl.w_gate.copy_(full.w_gate)
            epl.keys.copy_(full.keys[rank * El:(rank + 1) * El])
            epl.values.copy_(full.values[rank * El:(rank + 1) * El])
            if bias:
                epl.bias.copy_(full.bias[rank * El:(rank + 1) * El])
                epl.o_bias.copy_(full.o_bias)
        g = torch.Generator().manual_seed(70 + rank)
        x = torch.randn(B, N, D, generator=g)
        if empty_half:
            x[..., 0] = x[..., 0].abs() + 4.0
            with torch.no_grad():
                for lay in (full, epl):
                    lay.w_gate[: E // 2, 0] = 1.0
                    lay.w_gate[E // 2:, 0] = -1.0
        x = x.to(dev)
        dy = torch.randn(B, N, D, generator=g).to(dev)
        xa = x.clone().requires_grad_(True)
        xb = x.clone().requires_grad_(True)

        mod_a, mod_b = full, epl
        if block:
            from competesmoe_amd.pretrain import MoEBlock
            norms = [nn.LayerNorm(D).to(dev) for _ in range(2)]
            with torch.no_grad():
                for ln in norms:
                    ln.weight.copy_(torch.linspace(0.5, 1.5, D))
                    ln.bias.copy_(torch.linspace(-0.1, 0.1, D))
            mod_a, mod_b = MoEBlock(norms[0], full).train(), MoEBlock(norms[1], epl).train()

        def step(mod, layer, xin):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                out = mod(xin)
                reg = sum(layer.get_reg_loss().values())
            torch.autograd.backward([out, reg.float()], [dy.to(out.dtype), torch.ones((), device=dev)])
            return out, reg
        oa, ra = step(mod_a, full, xa)
        ob, rb = step(mod_b, epl, xb)
        torch.cuda.synchronize()
        ok = torch.equal(oa, ob) and float(ra) == float(rb)
        # fp32: every row goes through the same kernels -> the same bits.  bf16: d w is a sum of exact fp32 products whose grouping
        # follows the dH launch's tile class, chosen from the row count of the launch (T*K rows there, the received rows here): the
        # last fp32 bits of d w may differ, and with them a rounding of dx here and there
        if autocast:
            err = float((xa.grad - xb.grad).norm() / xa.grad.norm())
            ok = ok and err <= 1e-5
        else:
            ok = ok and torch.equal(xa.grad, xb.grad)
        tol = 1e-5 if not autocast else 3e-3
        names = ["keys", "values"] + (["bias"] if bias else [])
        for nme in names:
            gr = getattr(full, nme).grad.detach().float().cpu()
            if world > 1:
                dist.all_reduce(gr)
            gm = getattr(epl, nme).grad.detach().float().cpu()
            gr = gr[rank * El:(rank + 1) * El]
            ok = ok and float((gm - gr).norm() / (gr.norm() + 1e-12)) <= tol
        for nme in ["w_gate"] + (["o_bias"] if bias else []):
            gr = getattr(full, nme).grad.detach().float().cpu()
            if world > 1:
                dist.all_reduce(gr)
            ok = ok and float((getattr(epl, nme).grad.float().cpu() - gr).norm() / gr.norm()) <= tol
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _launch(world, backend, dt_name, chunks=1, empty_half=False, target=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target or _run, args=(r, world, port, backend, dt_name, q, chunks, empty_half)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return dict(q.get(timeout=5) for _ in range(world))


@pytest.mark.parametrize("dt_name", ["fp32", "bf16"])
def test_ep_world1_rccl_equals_single_gpu(dt_name):
    assert _launch(1, "nccl", dt_name) == {0: True}


@pytest.mark.parametrize("dt_name", ["fp32", "bf16"])
def test_ep_world2_one_gpu_equals_single_gpu(dt_name):
    assert _launch(2, "gloo", dt_name) == {0: True, 1: True}


@pytest.mark.parametrize("world,backend,dt_name", [(1, "nccl", "fp32"), (1, "nccl", "bf16"), (2, "gloo", "fp32")])
def test_competesmoe_ep_equals_single_gpu(world, backend, dt_name):
    """`competesmoe_ep`: a competition step (gather tokens, dense local experts, scatter affinities, sparse recompute through the
    exchange, diversity loss on the exchanged rows) and a router step give the single-GPU `competesmoe` layer's outputs bit for bit
    and its losses / gradients within accumulation-order noise."""
    assert _launch(world, backend, dt_name, target=_run_comp) == {r: True for r in range(world)}


@pytest.mark.parametrize("chunks", [2, 3, 8])
def test_ep_chunked_world1_rccl_equals_single_gpu(chunks):
    """The overlapped exchange (async list all-to-all on the RCCL process group's stream, groups of local experts): bit-identical
    to the single-GPU layer.  chunks = 3 gives uneven groups, chunks = 8 one expert per group."""
    assert _launch(1, "nccl", "bf16", chunks) == {0: True}


@pytest.mark.parametrize("dt_name,chunks", [("fp32", 2), ("bf16", 4)])
def test_ep_chunked_world2_one_gpu_equals_single_gpu(dt_name, chunks):
    assert _launch(2, "gloo", dt_name, chunks) == {0: True, 1: True}


@pytest.mark.parametrize("dt_name,chunks", [("bf16", 1), ("bf16", 2), ("fp32", 2)])
def test_ep_world4_one_gpu_equals_single_gpu(dt_name, chunks):
    """Four ranks (two local experts each) on the one GPU, collectives through gloo: the layer's full forward + backward with the
    HIP kernels at a world size beyond 2 -- outputs and dx bit-identical to the single-GPU layer, expert gradients equal to the
    all-reduced single-GPU ones, replicated gate gradient summed over the ranks.  (More ranks than this on one card are not allowed
    on the test box; the exchange plumbing alone runs at 8 ranks in tests/test_ep_gloo.py.)"""
    assert _launch(4, "gloo", dt_name, chunks) == {r: True for r in range(4)}


@pytest.mark.parametrize("world,backend,dt_name,chunks,empty_half",
                         [(1, "nccl", "bf16", 1, False), (1, "nccl", "bf16", 3, False), (2, "gloo", "fp32", 1, False),
                          (2, "gloo", "bf16", 2, False), (4, "gloo", "bf16", 2, False), (2, "gloo", "fp32", 2, True), (1, "nccl", "fp32", 2, True)])
def test_ep_direct_exchange_equals_single_gpu(world, backend, dt_name, chunks, empty_half, monkeypatch):
    """CSMOE_EP_DIRECT=1: one message per (peer, local expert) delivered expert-major (no regroup pass, no inverse before the return
    trip), plain and in overlapped groups, over RCCL at world 1 (self pair = device copies) and through gloo at worlds 2 and 4 on one
    GPU, with experts (and at world 2 a whole rank) that receive nothing: the same bits as the single-GPU layer."""
    monkeypatch.setenv("CSMOE_EP_DIRECT", "1")
    assert _launch(world, backend, dt_name, chunks, empty_half) == {r: True for r in range(world)}


@pytest.mark.parametrize("world,backend,chunks", [(1, "nccl", 1), (1, "nccl", 2), (2, "gloo", 1), (2, "gloo", 2)])
def test_ep_with_experts_that_receive_nothing(world, backend, chunks):
    """Half of the experts are never selected: empty groups of the overlapped exchange, zero-row grouped GEMMs, and (world 2) a
    rank that receives no rows at all and still takes part in every collective."""
    assert _launch(world, backend, "fp32", chunks, True) == {r: True for r in range(world)}


@pytest.mark.parametrize("world,backend,dt_name,chunks,empty_half,direct",
                         [(1, "nccl", "fp32", 1, False, False), (1, "nccl", "bf16", 1, False, False), (1, "nccl", "bf16", 3, False, True),
                          (1, "nccl", "bf16_bias", 2, False, False), (2, "gloo", "fp32", 1, False, False), (2, "gloo", "bf16", 2, False, False),
                          (2, "gloo", "bf16_bias", 2, False, True), (2, "gloo", "bf16", 2, True, False), (4, "gloo", "bf16", 2, False, True),
                          (2, "gloo", "fp32", 1, True, True), (1, "nccl", "bf16_block", 2, False, False), (2, "gloo", "bf16_block", 2, False, True),
                          (1, "nccl", "bf16_big", 2, False, False), (2, "gloo", "bf16_big", 1, False, True), (2, "gloo", "bf16", 4, False, False)])
def test_pretrain_smoe_ep_equals_single_gpu(world, backend, dt_name, chunks, empty_half, direct, monkeypatch):
    """Pretrain `smoe_ep` (packed experts sharded over the group, `ep.EPFFNPacked`): the single-GPU pretrain `smoe` layer's output and
    regulariser bit for bit, dx bit for bit in fp32 and to 1e-5 under bf16 autocast (see the worker), local expert gradients equal to
    the all-reduced single-GPU ones, replicated gate / o_bias gradients summed over the ranks -- plain and overlapped, per-peer and
    direct exchange, one expert per group (no regroup passes on the per-peer road), with experts (and at world 2 a whole rank) that
    receive nothing, and inside the pretrain block (fp32 residual stream
    added in the combine, logits from the fused LayerNorm + gate launch)."""
    monkeypatch.setenv("CSMOE_EP_DIRECT", "1" if direct else "0")
    assert _launch(world, backend, dt_name, chunks, empty_half, target=_run_pretrain) == {r: True for r in range(world)}


def test_bench_script_runs_small_config_and_ep_path():
    """bench.py end to end on a small layer: the single-GPU path and (with --force-ep, world size 1 over RCCL) the exact code
    path the driver launches for N > 1.  One JSON line with the contract's keys."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--tokens", "2048", "--seq", "512", "--d-model", "256", "--d-ff", "512",
            "--experts", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    for extra in ([], ["--force-ep"], ["--force-ep", "--ep-chunks", "2"], ["--force-ep", "--ep-trial"]):
        env = dict(os.environ, MASTER_PORT=str(_free_port()))
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline"):
            assert k in d, k
        assert d["value"] > 0 and d["n_gpus"] == 1 and d["config"]["workload"]
        if "--ep-trial" in extra:          # the N>1 control flow: trial of 1 / 2 / 4 groups and one expert per group (8), then the choice
            assert set(d["config"]["ep_chunks_trial_ms"]) == {"1", "2", "4", "8"} and d["config"]["ep_chunks"] in (1, 2, 4, 8)
        elif extra:
            assert ("ep_wait_exposed" if "--ep-chunks" in extra else "ep_all_to_all") in d["kernels"]
            assert d["config"]["ep_chunks"] == (2 if "--ep-chunks" in extra else 1)
