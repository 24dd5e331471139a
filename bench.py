#!/usr/bin/env python3
"""bench.py -- MoE-layer forward+backward tokens/s on MI355X (BASELINE.json metric).

One "step" = one forward + backward of ONE sparse-MoE layer (router -> bin -> dispatch -> grouped expert GEMMs ->
combine, and the whole backward incl. expert weight gradients) over one batch of synthetic tokens resident in HBM.

Workload at N=1 (BASELINE.json configs[1]): T=32768 tokens as [16, 2048], d_model=4096, d_ff=11008, 64 experts, top-2,
bf16, experts Linear(D,F)+b -> GELU -> Linear(F,D)+b, `smoe` routing; x ~ N(0,1) seed 0, gate N(0,0.02) generator-seed
42, expert weights N(0,0.02) seed 1, upstream gradient N(0,1) seed 2.
N>1: expert-parallel (competesmoe_amd.ep): experts sharded E/N per rank, T tokens PER RANK (weak scaling), RCCL
all-to-all over xGMI for dispatch/combine; launched by torch.distributed.run, one rank per GPU.

Prints ONE JSON line on rank 0 (see the driver contract): value = total tokens / max-over-ranks time of K steps.
`roofline` is for the dominant kernel (the grouped expert GEMM family): algorithmic FLOPs per launch / mean launch
duration from HIP events recorded on the launch stream inside the timed region.  `cpu_baseline` times the CPU oracle
(oracle/moe_oracle.py, a port of the reference layer) on a bounded sample on this host's cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0    # dense block-scaled fp8 peak, same guide
HBM_PEAK_GBS = 8000.0
ROUND = "r03"                    # profiles/<ROUND>/traffic.json holds the PMC passes of THIS round's kernels (see traffic_for)


# every source a grouped-GEMM launch of any workload can come from: the traffic figure is only quoted for the kernels it was measured on
GEMM_SOURCES = ("gemm_bf16_v4.hip", "gemm_bf16_v2.hip", "gemm_bf16_v2p.hip", "gemm_bf16_v2c.hip", "gemm_fp8.hip", "gemm_tiles.h",
                "gemm_epilogue.h", "common.h")


def kernel_sources_hash():
    """sha1 over the sources of the grouped GEMM kernels (bf16 v4 / v2 / wgrad, fp32-weight, MXFP8): ties a committed PMC traffic
    figure to the kernels it was measured on."""
    import hashlib
    h = hashlib.sha1()
    for f in GEMM_SOURCES:
        with open(os.path.join(ROOT, "competesmoe_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def workload_key(config, dtype):
    """What a traffic figure belongs to: the workload sentence of the JSON line (stack, routing, sizes, experts), the arithmetic
    type, the parallelism and, for the expert-parallel layer (also when forced onto one rank), its exchange form.  tools/summarize_profile.py writes it from the bench line printed under rocprofv3; traffic_for
    reads it back for the run being timed."""
    return {"workload": config["workload"], "dtype": dtype, "parallelism": config["parallelism"],
            "graph_replay": bool(config.get("graph_replay", False)),
            "ep": [config["ep_chunks"], bool(config.get("ep_direct", False))] if "ep_chunks" in config else None}


def traffic_for(kernel, config, dtype):
    """HBM-side bytes per launch of `kernel` from this round's committed rocprofv3 --pmc passes (PMC counters cannot be read from
    inside this process), ONLY if they were collected on this workload (workload_key) with the kernels being timed now (matching
    source hash); else None and the reason."""
    try:
        with open(os.path.join(ROOT, "profiles", ROUND, "traffic.json")) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None, f"no profiles/{ROUND}/traffic.json"
    key = workload_key(config, dtype)
    for name, w in d.get("workloads", {}).items():
        if w.get("key") == key:
            if w.get("kernel_sources_sha1_16") != kernel_sources_hash():
                return None, f"profiles/{ROUND}/traffic.json[{name}] was collected on other kernel sources: not reported"
            return w.get("bytes_per_launch", {}).get(kernel), f"profiles/{ROUND}/traffic.json[{name}]: {w.get('note', '')}"
    return None, f"profiles/{ROUND}/traffic.json holds no PMC pass of this workload: not reported"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # warm-up 10, measure >= 50 iterations
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tokens", type=int, default=32768)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--d-model", type=int, default=4096)
    ap.add_argument("--d-ff", type=int, default=11008)
    ap.add_argument("--experts", type=int, default=64)
    ap.add_argument("--topk", type=int, default=2)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="fp8: BASELINE config 5 -- the pretrain stack's deepseekv2 layer (routed experts + --shared shared experts) with "
                         "GEMM 1 / GEMM 2 / dH / dXs on the MXFP8 matrix pipe (fp32 master weights quantised directly, bf16 autocast)")
    ap.add_argument("--weight-cache", action="store_true",
                    help="--stack pretrain (bf16): keep the bf16 operand copies of the fp32 master weights while the parameters are unchanged "
                         "(functional.weight_cache; the timed steps then model the micro-batches after the first of a gradient-accumulation "
                         "step).  Off by default: every step converts the masters")
    ap.add_argument("--fp8-weight-cache", action="store_true",
                    help="--dtype fp8: keep the quantised expert weights while the parameters are unchanged (args.fp8_weight_cache; the timed "
                         "steps then model the micro-batches after the first of a gradient-accumulation step).  Off by default: every step quantises")
    ap.add_argument("--shared", type=int, default=0, help="number of shared experts (width n x d_ff) of the deepseekv2 layer (fp8 / --stack pretrain)")
    ap.add_argument("--skew", action="store_true", help="add +2.0 to 8 gate rows (Zipf-like load, BASELINE.md)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=4096)
    ap.add_argument("--graph", action="store_true", help="time REPLAYS of the step captured as one hipGraph (competesmoe_amd.graphs.GraphedStep): the launch-bound small shapes; the per-kernel table comes from eager steps run before the capture")
    ap.add_argument("--competition", action="store_true", help="time the CompeteSMoE competition step (every expert dense + sparse recompute) instead of the sparse smoe step")
    ap.add_argument("--block", action="store_true", help="time the block around the layer, x + MoE(LayerNorm(x)) (SURVEY.md section 8 f1), with the fused LayerNorm+gate / residual-combine kernels")
    ap.add_argument("--block-unfused", action="store_true", help="same block composed from torch LayerNorm, the plain layer and a torch add (A/B for --block)")
    ap.add_argument("--stack", default="llava", choices=["llava", "pretrain"], help="pretrain: the LM-pretrain stack's `smoe` layer (packed fp32 master weights keys/values, ReLU, no bias, bf16 autocast: the cvmm path) instead of the LLaVA-stack layer")
    ap.add_argument("--ep-chunks", type=int, default=0, help="expert-parallel runs: groups of local experts whose all-to-all overlaps the grouped GEMMs (competesmoe_amd.ep); 0 = pick the fastest of 1 / 2 / 4 / one expert per group in a short untimed trial before the warmup (1 with a single rank)")
    ap.add_argument("--ep-trial", action="store_true", help="run the overlap-depth trial even with a single rank (exercises the N>1 control flow on one GPU)")
    ap.add_argument("--stub", action="store_true", help="launcher test: CPU stand-in step over gloo, no GPU (tests/test_bench_launcher.py)")
    ap.add_argument("--ep-direct", action="store_true", help="expert-parallel runs: one message per (peer, local expert) delivered expert-major "
                    "(no regroup passes; EPSMoeLayer(direct=True)).  Off by default, see DESIGN section 5")
    ap.add_argument("--force-ep", action="store_true", help="use the expert-parallel layer even with one rank (smoke-tests the N>1 code path)")
    return ap.parse_args()


def make_layer(a, dev, dt, E_local=None, seed=1):
    from competesmoe_amd.moe import get_moe
    D, F, E, K = a.d_model, a.d_ff, a.experts, a.topk
    args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001)
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.device(dev):
        experts = nn.ModuleList([nn.Sequential(nn.Linear(D, F, dtype=dt), nn.GELU(), nn.Linear(F, D, dtype=dt))
                                 for _ in range(E if E_local is None else E_local)])
    with torch.no_grad():
        for m in experts:
            for p in m.parameters():
                p.normal_(0.0, 0.02, generator=g)
    if E_local is None and a.competition:
        args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001, rate_flip=1.0, warm_up=0.0,
                                     max_compete_in_iter=8, router_loss_coef=0.01, diversity_loss_coef=0.01,
                                     bal_comp_loss_coef=0.01, hybrid=False, router_theta=0.5, moe_name="competesmoe")
        layer = get_moe("competesmoe")(D, D, E, K, experts, args)
        layer.set_total_steps(16, 0, {})
        layer.prob_flips = torch.ones_like(layer.prob_flips)      # compete on every step
        layer._flips_host = None
        layer.set_current_steps(1)
    elif E_local is None:
        layer = get_moe("smoe")(D, D, E, K, experts, args)
    else:
        from competesmoe_amd.ep import EPSMoeLayer
        layer = EPSMoeLayer(D, D, E, K, experts, args)
    layer = layer.to(dev).to(dt).train()
    if a.skew:
        with torch.no_grad():
            layer.gate.weight[:8] += 2.0 / (D ** 0.5)
    return layer


def make_pretrain_layer(a, dev, ep=False):
    """pretrain-stack `smoe` (moe_pretrain_model/layers/moe/smoe.py): fp32 master parameters, the step runs under bf16 autocast.
    `ep`: the expert-parallel form `smoe_ep` (this rank's E/P experts of the packed tensors; competesmoe_amd/pretrain/smoe_ep.py)."""
    import torch.nn.functional as F
    from competesmoe_amd.pretrain import get_moe
    args = types.SimpleNamespace(moe_name="smoe", stop_after=10, warm_up=0.0, rate_flip=1.0, max_compete_in_iter=8,
                                 balance_loss_coef=0.01, balance_loss_coef_comp=0.02, router_loss_coef=0.03, router_theta=0.5,
                                 in_topk=False, hybrid=False, tribrid=False, balance_affinity=False, is_cosine=False,
                                 is_norm_weight=False, norm_sigmoid=False, scale_weight=1.0, test_only=False)
    name = "smoe_ep" if ep else "smoe"
    if a.dtype == "fp8" or a.shared > 0:
        name = "deepseekv2" if a.shared > 0 else "smoe"
        args.fp8_experts = a.dtype == "fp8"
        args.fp8_weight_cache = bool(a.fp8_weight_cache)
        args.n_shared_experts = max(1, a.shared)
    with torch.device(dev):
        layer = get_moe(name)(a.d_model, a.experts, a.d_ff, n_heads=a.topk, activation=F.relu, bias=False, log_interval=None,
                              args=args)
    layer = layer.to(dev).train()
    layer.regularization_present = True
    return layer


def host_cores():
    """(physical cores, logical CPUs, model name) from /proc/cpuinfo: distinct (physical id, core id) pairs."""
    pairs, logical, model, phys_id = set(), 0, "unknown", "0"
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("processor"):
                    logical += 1
                elif ln.startswith("model name") and model == "unknown":
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    phys_id = ln.split(":", 1)[1].strip()
                elif ln.startswith("core id"):
                    pairs.add((phys_id, ln.split(":", 1)[1].strip()))
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    return (len(pairs) or logical), logical, model


def cpu_baseline(a):
    """CPU oracle (port of the reference layer) on a bounded sample: same D/F/E/K, fp32, `cpu_tokens` tokens."""
    from oracle import moe_oracle as O
    D, F, E, K, T = a.d_model, a.d_ff, a.experts, a.topk, a.cpu_tokens
    phys, logical, cpu_model = host_cores()
    cores = max(1, min(phys, len(os.sched_getaffinity(0))))     # one thread per PHYSICAL core this process may run on
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1)
    w1 = torch.empty(F, D).normal_(0, 0.02, generator=g)
    w2 = torch.empty(D, F).normal_(0, 0.02, generator=g)
    experts = []
    for _ in range(E):   # clones of one random draw: same arithmetic/traffic as independent weights, faster to set up
        experts.append(tuple(t.requires_grad_(True) for t in (w1.clone(), torch.zeros(F), w2.clone(), torch.zeros(D))))
    wg = torch.empty(E, D).normal_(0, 0.02, generator=torch.Generator().manual_seed(42)).requires_grad_(True)
    x = torch.randn(1, T, D, generator=torch.Generator().manual_seed(0)).requires_grad_(True)
    dy = torch.randn(1, T, D, generator=torch.Generator().manual_seed(2))
    args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001)
    t0 = time.perf_counter()
    out, aux, _, _ = O.llava_smoe_forward(x, wg, experts, "gelu", K, args)
    torch.autograd.backward([out, aux], [dy, torch.ones(())])
    dt_s = time.perf_counter() - t0
    # BASELINE config 1 in full (the reference's own CPU-runnable case): pretrain-style layer D=256, E=8, K=2, F=128, 1024 tokens
    D1, F1, E1, K1, T1 = 256, 128, 8, 2, 1024
    g1 = torch.Generator().manual_seed(0)
    x1 = torch.randn(4, 256, D1, generator=g1).requires_grad_(True)
    k1 = (torch.randn(E1, D1, F1, generator=g1) * D1 ** -0.5).requires_grad_(True)
    v1 = (torch.randn(E1, F1, D1, generator=g1) * (E1 * F1) ** -0.5).requires_grad_(True)
    wg1 = (torch.randn(E1, D1, generator=g1) * D1 ** -0.5).requires_grad_(True)
    # a 1024-token problem: 8 threads (the reference's own CPU-runnable case is a workstation-sized one; on 64+ threads the
    # per-op fork/join of torch's pool costs more than the 0.2 GFLOP of work and the timing swung 12x between boxes in round 2),
    # two untimed passes after the 23-GB job above, then the MEDIAN of 7
    c1 = min(8, cores)
    torch.set_num_threads(c1)
    times = []
    for it in range(9):
        for t_ in (x1, k1, v1, wg1):
            t_.grad = None
        t1 = time.perf_counter()
        lg = O.gate_logits(x1, wg1)
        w1_, i1, _ = O.router_topk(lg, K1, x1.dtype)
        o1 = O.pretrain_ffn(x1, i1, w1_, k1, v1, "relu", torch.float32)
        (o1.sum() + O.entropy_balance(lg) * 0.01).backward()
        if it >= 2:
            times.append(time.perf_counter() - t1)
    torch.set_num_threads(cores)
    med = sorted(times)[len(times) // 2]
    return {"value": T / dt_s, "unit": "tokens/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "physical_cores": phys, "logical_cpus": logical,
            "config1_tokens_per_s": round(T1 / med, 1), "config1_threads": c1,
            "sample": f"CPU oracle (fp32 port of the reference SMoE layer) on {cores} threads = one per physical core of {cpu_model} "
                      f"({phys} physical cores, {logical} logical CPUs): one fwd+bwd of {T} tokens, "
                      f"D={D} F={F} E={E} K={K}, {dt_s:.1f} s; BASELINE config 1 in full (pretrain-style layer D=256 F=128 E=8 K=2, "
                      f"1024 tokens) on {c1} threads: median of 7 after 2 warm passes {med * 1e3:.1f} ms per fwd+bwd "
                      f"(min {min(times) * 1e3:.1f}, max {max(times) * 1e3:.1f})"}


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python3 bench.py --gpus N` typed as is (no launcher around it): start N fresh ranks -- `python -m torch.distributed.run`,
    one process per GPU, rendezvous on 127.0.0.1 -- as a CHILD of this process, which has not touched the GPU (importing torch does
    not initialise HIP), relay their output (rank 0 prints the ONE JSON line) and return the children's exit status.  Never an
    exec: a process that has initialised the GPU must not be replaced, and this one stays alive to pass the status on.
    (The reference's launch for comparison: moe_pretrain_model/train.sh:34-39.)"""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what this pool's driver supports (RCCL over xGMI)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def stub_main(a, rank, world):
    """--stub: the N>1 control flow (rendezvous, barrier-bracketed timed region, max over ranks, one JSON line from rank 0) over
    gloo with a CPU stand-in for the step -- what tests/test_bench_launcher.py runs at world 2 in a container without GPUs.  Not a
    measurement: `data` says so."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w = torch.randn(64, 64)

    def step():
        return (w @ w).sum()

    def fence():
        if world > 1:
            dist.barrier()
    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    el = float(tt)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": round(a.tokens * world * a.steps / el, 1), "unit": "tokens/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "stub step on CPU over gloo (launcher test)",
                          "config": {"workload": "launcher stub", "ranks": world, "comm_world_size": dist.get_world_size() if world > 1 else 1}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} ranks", file=sys.stderr)
        sys.exit(2)
    if a.stub:
        return stub_main(a, rank, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.dtype == "fp8":
        a.stack = "pretrain"
    dt = torch.float32 if a.dtype == "fp32" else torch.bfloat16
    import torch.distributed as dist
    if world > 1 or a.force_ep:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from competesmoe_amd import ops
    T, D = a.tokens, a.d_model
    Bsz = max(1, T // a.seq)
    Nseq = T // Bsz
    if a.stack == "pretrain":
        ep_pre = world > 1 or a.force_ep
        assert not a.competition and not (ep_pre and (a.dtype == "fp8" or a.shared > 0)), \
            "--stack pretrain: the smoe step (optionally --block; expert-parallel over the ranks), or the single-GPU fp8 / shared-expert layer"
        assert a.experts % world == 0, "experts must divide over ranks"
        layer = make_pretrain_layer(a, dev, ep=ep_pre)
        if a.weight_cache:
            from competesmoe_amd import functional as Fn
            Fn.weight_cache(True)
    elif world > 1 or a.force_ep:
        assert a.experts % world == 0, "experts must divide over ranks"
        layer = make_layer(a, dev, dt, E_local=a.experts // world, seed=1 + rank)
    else:
        layer = make_layer(a, dev, dt)
    blk = ln = None
    if (a.block or a.block_unfused) and a.stack == "pretrain":
        ln = nn.LayerNorm(D).to(dev)                       # fp32 parameters, fp32 residual stream (relative_moe_transformer.py:128-129)
        if a.block:
            from competesmoe_amd.pretrain import MoEBlock as PretrainBlock
            blk = PretrainBlock(ln, layer, 0.0).train()
    elif a.block or a.block_unfused:
        ln = nn.LayerNorm(D, eps=1e-6).to(dev).to(dt)
        if a.block:
            from competesmoe_amd.moe import MoEBlock
            blk = MoEBlock(ln, layer)
    x = torch.randn(Bsz, Nseq, D, device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(rank)).to(dt)
    dy = torch.randn(Bsz, Nseq, D, device=dev, dtype=torch.float32, generator=torch.Generator(device=dev).manual_seed(2 + rank)).to(dt)
    x.requires_grad_(True)
    one = torch.ones((), device=dev)
    x32 = dy32 = None
    dy_as = {}
    if a.stack == "pretrain":          # the LM's residual stream is fp32 under autocast
        x32 = x.detach().float().requires_grad_(True)
        dy32 = dy.float()

    def step():
        for p in layer.parameters():
            p.grad = None
        x.grad = None
        if a.stack == "pretrain":
            if ln is not None:
                ln.weight.grad = ln.bias.grad = None
            x32.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                if blk is not None:
                    out = blk(x32, id_layer=0)
                elif ln is not None:
                    out = x32 + layer(ln(x32), id_layer=0)
                else:
                    out = layer(x32, id_layer=0)
                reg = sum(layer.get_reg_loss().values())
            # the upstream gradient in the output's dtype (bf16 under autocast, as the LLaVA stack's dy): resident input, cast once
            g = dy_as.get(out.dtype)
            if g is None:
                g = dy_as[out.dtype] = dy32.to(out.dtype)
            torch.autograd.backward([out, reg.float()], [g, one])
            return
        if blk is not None:
            out, aux, _, _ = blk(x)
        elif ln is not None:
            ln.weight.grad = ln.bias.grad = None
            out, aux, _, _ = layer(ln(x))
            out = x + out
        else:
            out, aux, _, _ = layer(x)
        torch.autograd.backward([out, aux.float()], [dy, one])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ep_tune = None
    if world > 1 or a.force_ep:
        layer.direct = bool(a.ep_direct)
        if a.ep_chunks > 0 or (world == 1 and not a.ep_trial):
            layer.chunks = max(1, a.ep_chunks)
        else:   # untimed trial: overlap depth that is fastest on THIS node (max over ranks, so every rank picks the same)
            ep_tune = {}
            El = a.experts // world
            for c in sorted({1, 2, 4, El} if El <= 16 else {1, 2, 4}):     # El groups = one expert each: no regroup passes
                if c > El:
                    continue
                layer.chunks = c
                step()
                fence()
                t0 = time.perf_counter()
                for _ in range(3):
                    step()
                fence()
                tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ep_tune[c] = round(float(tt) / 3 * 1e3, 3)
            layer.chunks = min(ep_tune, key=ep_tune.get)
    for _ in range(a.warmup):
        step()
    fence()
    if a.graph:
        assert world == 1 and a.stack == "llava" and blk is None and ln is None and not a.force_ep, "--graph: single-GPU LLaVA-stack layer only"
        from competesmoe_amd.graphs import GraphedStep
        ops.profile_start()
        for _ in range(min(5, a.steps)):
            step()
        fence()
        prof = ops.profile_stop()
        prof = {k: {**v, "calls": v["calls"] * a.steps / min(5, a.steps)} for k, v in prof.items()}       # per-step counts of the table below
        x.grad = None
        for p in layer.parameters():
            p.grad = None

        def loss_fn(xi):
            out, aux, _, _ = layer(xi)
            return (out.float() * dy.float()).sum() + aux.float()
        gstep = GraphedStep(loss_fn, [x], list(layer.parameters()))
        for _ in range(a.warmup):
            gstep(x)
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            gstep(x)
        fence()
        el = time.perf_counter() - t0
    else:
        ops.profile_start()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        el = time.perf_counter() - t0
        prof = ops.profile_stop()
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt)
    ms = el / a.steps * 1e3
    total_tokens = Bsz * Nseq * world

    if rank == 0:
        # ---- roofline of the dominant kernel family (grouped expert GEMM): FLOPs per launch / mean launch duration
        gemm = {k: v for k, v in prof.items() if k.startswith("grouped_gemm") or k.startswith("grouped_wgrad") or k.startswith("dense_wgrad") or k == "gate_wgrad"
                or k.startswith("dense_gemm")}
        detail = {}
        for k, v in prof.items():
            if k in gemm:
                detail[k] = {"calls_per_step": v["calls"] / a.steps, "ms": round(v["ms"], 4),
                             "TFLOP/s": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 1)}
            else:
                detail[k] = {"calls_per_step": v["calls"] / a.steps, "ms": round(v["ms"], 4),
                             "GB/s": round(v["work"] / (v["ms"] * 1e-3) / 1e9, 1)}
        roof = None
        if gemm:
            dom = max((k for k in gemm if k != "gate_wgrad"), key=lambda k: gemm[k]["ms"] * gemm[k]["calls"])
            ach = gemm[dom]["work"] / (gemm[dom]["ms"] * 1e-3) / 1e12
            def peak_of(k):      # the matrix-pipe peak of the dtype THAT kernel computes in
                if "mxfp8" in k:
                    return MFMA_FP8_PEAK_TFLOPS
                return 157.3 if a.dtype == "fp32" else MFMA_BF16_PEAK_TFLOPS
            peak = peak_of(dom)
            big = {k: v for k, v in gemm.items() if k != "gate_wgrad"}
            tot_ms = sum(v["ms"] * v["calls"] for v in big.values()) / a.steps
            tot_fl = sum(v["work"] * v["calls"] for v in big.values()) / a.steps
            tot_peak_ms = sum(v["work"] * v["calls"] / (peak_of(k) * 1e12) for k, v in big.items()) / a.steps * 1e3
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": None, "traffic_note": None,
                    "all_grouped_gemm": {"ms_per_step": round(tot_ms, 3), "TFLOP/s": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1),
                                         "frac": round(tot_peak_ms / tot_ms, 4)},
                    "hbm_kernels": {k: {"GB/s": d["GB/s"], "frac": round(d["GB/s"] / HBM_PEAK_GBS, 4)}
                                    for k, d in detail.items() if "GB/s" in d}}
        res = {
            "metric": f"MoE-layer fwd+bwd tokens/sec at d_model={D}, {a.experts} experts top-{a.topk}", "value": round(total_tokens * a.steps / el, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ((f"BASELINE config 5: pretrain-stack deepseekv2 layer, {a.experts} routed + {a.shared} shared experts, GEMM 1 / GEMM 2 / dH / dXs on the MXFP8 matrix pipe (fp32 master weights quantised directly, bf16 weight gradients" + ("; quantised weights reused while the parameters are unchanged, as over the micro-batches of one optimizer step" if a.fp8_weight_cache else "; weights quantised every step") + "): " if a.dtype == "fp8" else ("pretrain-stack layer (packed fp32 master weights, ReLU, no bias, bf16 autocast; bf16 operand copies reused while the parameters are unchanged, as over the micro-batches of one optimizer step): " if a.weight_cache else "pretrain-stack layer (packed fp32 master weights, ReLU, no bias, bf16 autocast; weights cast to bf16 every step): ")) if a.stack == "pretrain" else "") + ("block x + MoE(LayerNorm(x)) " + ("(fused LayerNorm+gate, residual in combine) around a " if a.block else "(unfused: torch LayerNorm + add) around a ") if (a.block or a.block_unfused) else "") + f"single sparse-MoE layer ({'competesmoe competition step' if a.competition else 'smoe routing'}), T={Bsz * Nseq} tokens/GPU as [{Bsz},{Nseq}], "
                                   f"d_model={D}, d_ff={a.d_ff}, {a.experts} experts top-{a.topk}, " + ("ReLU experts without bias, " if a.stack == "pretrain" else "Linear+bias/GELU experts, ") +
                                   f"fwd+bwd incl. expert weight grads" + (", skewed gate" if a.skew else ""),
                       "graph_replay": bool(a.graph), "tokens_per_gpu": Bsz * Nseq, "d_model": D, "d_ff": a.d_ff, "experts": a.experts, "top_k": a.topk,
                       "parallelism": "single GPU" if world == 1 else f"ep{world} (experts sharded, RCCL all-to-all)",
                       **({"ep_chunks": layer.chunks, "ep_direct": bool(a.ep_direct), "ep_chunks_trial_ms": ep_tune, "comm_world_size": dist.get_world_size(),
                           "comm_backend": "nccl (RCCL)"} if (world > 1 or a.force_ep) else {})},
            "roofline": roof, "kernels": detail,
            "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
        }
        if roof is not None:
            roof["traffic"], roof["traffic_note"] = traffic_for(roof["kernel"], res["config"], a.dtype)
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(a)
            except Exception as e:   # the GPU number stands even if the host cannot hold the CPU sample
                res["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(res))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
