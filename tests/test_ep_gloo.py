"""CPU, world_size 2, gloo: the expert-parallel exchange plumbing (plan, variable-size all-to-all, local regrouping)
reproduces the single-process result.  The HIP kernels between the exchanges are replaced here by a stand-in per-row
'expert' (row * (global_expert_id + 1)) -- the N>1 compute itself is covered on the GPU (tests/test_ep_gpu.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, D, E, K, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from competesmoe_amd import ep
        from oracle import moe_oracle as O
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(T, D, generator=g)
        sc = torch.rand(T, E, generator=g)
        if rank == 0:
            sc[:, 0] += 1.0            # skew: rank 0's tokens prefer expert 0
        sc[:, E - 1] = -1.0            # last expert never selected (empty bin)
        idx = sc.topk(K, -1).indices.int()
        counts, offsets, perm = O.bin_tokens(idx, E)
        xs = x[(perm // K)]
        plan = ep.make_plan(counts.int(), None)
        El = E // world
        assert plan.send_splits == [int(counts[p * El:(p + 1) * El].sum()) for p in range(world)]
        recv = ep.a2a_rows(xs, plan.send_splits, plan.recv_splits)
        ids = ep.local_expert_ids(plan)
        assert recv.shape[0] == plan.R == ids.numel()
        # stand-in expert: scale by (global expert id + 1)
        gid = ids.long() + rank * El
        y_recv = recv * (gid + 1).unsqueeze(1).float()
        y = ep.a2a_rows(y_recv, plan.recv_splits, plan.send_splits)
        # single-process expectation in the local binned order
        e_sorted = idx.flatten()[perm].long()
        exp = xs * (e_sorted + 1).unsqueeze(1).float()
        ok = torch.equal(y, exp)
        # chunked exchange (groups of local experts, per-peer row views): same rows back, group by group
        plan2 = ep.make_plan(counts.int(), None, per_expert=True)
        assert plan2.send_splits == plan.send_splits and plan2.recv_splits == plan.recv_splits and plan2.R == plan.R
        for chunks in (1, 2, 3, El):
            cps = ep.chunk_plan(plan2, chunks)
            assert [c.e0 for c in cps][0] == 0 and cps[-1].e1 == El and sum(c.R for c in cps) == plan.R
            y2 = torch.full_like(xs, float("nan"))
            for cp in cps:
                r = torch.empty(cp.R, D)
                ep.exchange_views(ep._packed_views(r, cp.recv_n), ep._views(xs, cp.send_lo, cp.send_n)).wait()
                ids_c = ep.local_expert_ids(plan2, cp.e0, cp.e1)
                assert ids_c.numel() == cp.R and (cp.R == 0 or int(ids_c.max()) < cp.e1 - cp.e0)
                gid_c = ids_c.long() + cp.e0 + rank * El
                ret = r * (gid_c + 1).unsqueeze(1).float()
                ep.exchange_views(ep._views(y2, cp.send_lo, cp.send_n), ep._packed_views(ret, cp.recv_n)).wait()
            ok = ok and torch.equal(y2, exp)
        # direct exchange: one message per (peer, local expert) straight into an expert-major buffer -- the rows the regroup pass
        # would have produced (stable by source rank inside an expert), offsets from the counts, and the way back; whole and in groups
        want = recv[torch.sort(ids.long(), stable=True).indices]
        for chunks in (1, 2, El):
            cps = ep.chunk_plan(plan2, chunks)
            y3 = torch.full_like(xs, float("nan"))
            got = []
            for cp in cps:
                r = torch.empty(cp.R, D)
                ep.exchange_direct(ep.direct_views(r, plan2, cp.e0, cp.e1, True), ep.direct_views(xs, plan2, cp.e0, cp.e1, False)).wait()
                lb = ep.local_bins(plan2, cp.e0, cp.e1)
                off = lb.offsets.long()
                assert int(off[-1]) == cp.R == lb.n and torch.equal(off[1:] - off[:-1], plan2.recv_counts[:, cp.e0:cp.e1].sum(0).long())
                got.append(r)
                gid_c = torch.repeat_interleave(torch.arange(cp.e1 - cp.e0), off[1:] - off[:-1]) + cp.e0 + rank * El
                ret = r * (gid_c + 1).unsqueeze(1).float()
                ep.exchange_direct(ep.direct_views(y3, plan2, cp.e0, cp.e1, False), ep.direct_views(ret, plan2, cp.e0, cp.e1, True)).wait()
            ok = ok and torch.equal(torch.cat(got), want) and torch.equal(y3, exp)
        # the lanes of the packed (pretrain) exchange, ep._Lane: wide rows on the direct road, and the narrow fp32 column (routing
        # weights out, dot products back) on both roads -- the per-peer road regroups a column by torch indexing with the lane's
        # local binning (built by the HIP binning kernel in the product; by the oracle's here)
        from competesmoe_amd import ops
        col = torch.arange(xs.shape[0], dtype=torch.float32).view(-1, 1) + 1000.0 * rank
        for direct in (True, False):
            for chunks in (1, 2, El):
                y4 = torch.full_like(xs, float("nan"))
                c4 = torch.full_like(col, float("nan"))
                for cp in ep.chunk_plan(plan2, chunks):
                    ln = ep._Lane(plan2, cp, direct, None)
                    Ec = cp.e1 - cp.e0
                    if not direct:
                        cn, of, pm = O.bin_tokens(ep.local_expert_ids(plan2, cp.e0, cp.e1).view(-1, 1), Ec)
                        so = torch.empty_like(pm)
                        so[pm.long()] = torch.arange(pm.numel(), dtype=pm.dtype)
                        ln._lb = ops.Bins(cn, of, pm, so, int(pm.numel()), Ec, 1)
                    off = ln.lb.offsets.long()
                    gid_c = (torch.repeat_interleave(torch.arange(Ec), off[1:] - off[:-1]) + cp.e0 + rank * El + 1).view(-1, 1).float()
                    rc, wk = ln.send(col)
                    wk.wait()
                    wk, _keep = ln.give_back(ln.arrived(rc) * gid_c, c4)
                    wk.wait()
                    if direct:
                        rr, wk = ln.send(xs)
                        wk.wait()
                        wk, _keep = ln.give_back(ln.arrived(rr) * gid_c, y4)
                        wk.wait()
                ok = ok and torch.equal(c4, col * (e_sorted + 1).view(-1, 1).float()) and (not direct or torch.equal(y4, exp))
        # all ranks' counts line up: recv_counts[s] == rank s's send counts for my experts
        allc = [torch.zeros(E, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(allc, counts.int())
        for s in range(world):
            ok = ok and torch.equal(plan.recv_counts[s], allc[s][rank * El:(rank + 1) * El])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("T,D,E,K", [(64, 8, 8, 2), (33, 4, 4, 1), (128, 16, 16, 3)])
def test_ep_exchange_two_ranks_gloo(T, D, E, K):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, D, E, K, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}


@pytest.mark.parametrize("world", [4, 8])
def test_ep_exchange_at_the_node_sizes_of_the_scaling_run_gloo(world):
    """The same plumbing at 4 and 8 ranks (the sizes the driver's scaling run uses; no 8-GPU node is available to the build): plan,
    variable-size exchange, chunked per-peer views, every rank's counts lining up -- with 2 local experts per rank at 8 ranks and an
    expert nobody selects."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 96, 8, 16, 2, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert res == {r: True for r in range(world)}


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from competesmoe_amd import ep
        w = torch.nn.Parameter(torch.arange(12, dtype=torch.float32).view(3, 4) / 10)
        ep.reduce_grad_on_backward(w, None)
        xs = [torch.randn(5, 4, generator=torch.Generator().manual_seed(10 * mb + r)) for r in range(world) for mb in range(2)]
        mine = [xs[rank * 2 + mb] for mb in range(2)]
        for x in mine:                                   # two micro-batches accumulate into w.grad
            (x @ w.t()).square().sum().backward()
        # expectation: ONE backward over every rank's and micro-batch's rows
        w2 = w.detach().clone().requires_grad_(True)
        (torch.cat(xs) @ w2.t()).square().sum().backward()
        q.put((rank, bool(torch.allclose(w.grad, w2.grad, rtol=1e-5, atol=1e-6))))
    finally:
        dist.destroy_process_group()


def test_replicated_gate_gradient_with_two_micro_batches_gloo():
    """the gate gradient of the expert-parallel layer is reduced per backward pass, so gradient accumulation over
    micro-batches gives sum_r sum_mb g and not P * (earlier sums) + ..."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}


def _comp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from competesmoe_amd import ep
        T, D, El = 6, 5, 3
        E = El * world
        xs = [torch.randn(T, D, generator=torch.Generator().manual_seed(40 + r)) for r in range(world)]
        Ws = [torch.randn(D, generator=torch.Generator().manual_seed(90 + e)) for e in range(E)]       # stand-in "expert" e: <x, W_e>
        x = xs[rank].clone().requires_grad_(True)
        mine = [Ws[rank * El + i].clone().requires_grad_(True) for i in range(El)]
        xa = ep.AllGatherRows.apply(x, None)                                         # [P*T, D]
        ok = torch.equal(xa.detach(), torch.cat(xs))
        aff_local = torch.stack([torch.tanh(xa @ w) for w in mine], dim=-1)          # [P*T, El]: all tokens x my experts
        aff = ep.ScatterAffinities.apply(aff_local, None)                            # [T, E]: my tokens x all experts
        want = torch.stack([torch.tanh(xs[rank] @ w) for w in Ws], dim=-1)
        ok = ok and torch.allclose(aff.detach(), want, atol=1e-6)
        coef = torch.arange(1, E + 1, dtype=torch.float32) * (rank + 1)              # a loss that differs per rank and per expert
        (aff * coef).sum().backward()
        # single-process expectation: total loss = sum_r sum_e coef_r[e] * tanh(<x_r, W_e>)
        xr = [t.clone().requires_grad_(True) for t in xs]
        wr = [w.clone().requires_grad_(True) for w in Ws]
        tot = sum((torch.stack([torch.tanh(xr[r] @ w) for w in wr], -1) * (torch.arange(1, E + 1, dtype=torch.float32) * (r + 1))).sum()
                  for r in range(world))
        tot.backward()
        ok = ok and torch.allclose(x.grad, xr[rank].grad, atol=1e-5)                # token gradients come home summed over the ranks' experts
        for i in range(El):                                                          # expert gradients are complete on the owner
            ok = ok and torch.allclose(mine[i].grad, wr[rank * El + i].grad, atol=1e-5)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_competition_exchanges_two_ranks_gloo():
    """The two exchanges of the expert-parallel competition step (competesmoe_ep): gather every rank's tokens, scatter every token's
    affinities back to its owner; values and both gradients against the single-process computation."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}


def _pretrain_ctor_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types
        import torch.nn.functional as F
        from competesmoe_amd.pretrain import get_moe
        args = types.SimpleNamespace(balance_loss_coef=0.01)
        lay = get_moe("smoe_ep")(32, 8, 16, n_heads=2, activation=F.relu, bias=True, args=args)
        ok = tuple(lay.w_gate.shape) == (8, 32) and tuple(lay.keys.shape) == (8 // world, 32, 16)
        ok = ok and tuple(lay.values.shape) == (8 // world, 16, 32) and tuple(lay.bias.shape) == (8 // world, 16)
        ok = ok and set(lay.state_dict()) == {"w_gate", "keys", "values", "bias", "o_bias"} and lay.n_experts == 8
        try:
            get_moe("smoe_ep")(32, 9, 16, n_heads=2, activation=F.relu, args=args)       # 9 experts do not divide over the ranks
            ok = ok and world == 1
        except ValueError:
            pass
        # the replicated gate's gradient is summed over the group once per backward pass
        (lay.w_gate * (rank + 1.0)).sum().backward()
        ok = ok and torch.equal(lay.w_gate.grad, torch.full_like(lay.w_gate, float(sum(r + 1 for r in range(world)))))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_pretrain_smoe_ep_holds_its_share_of_the_packed_experts_gloo():
    """Pretrain `smoe_ep` (pretrain/smoe_ep.py) at world 2 on the CPU: the constructor takes the arguments of `smoe`, holds E/P experts
    of the packed tensors under the reference's parameter names, refuses an expert count the ranks do not divide, and sums the
    replicated gate's gradient over the group.  (Its forward / backward against the single-GPU layer: tests/test_ep_gpu.py.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pretrain_ctor_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(2)) == {0: True, 1: True}


def test_pretrain_smoe_ep_needs_a_process_group():
    import types
    import torch.nn.functional as F
    from competesmoe_amd.pretrain import get_moe
    with pytest.raises(RuntimeError, match="process group"):
        get_moe("smoe_ep")(32, 8, 16, n_heads=2, activation=F.relu, args=types.SimpleNamespace(balance_loss_coef=0.01))
