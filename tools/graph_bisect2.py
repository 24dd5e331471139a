#!/usr/bin/env python3
"""Second bisect of the capture crash: variants of the failing test, one child process each."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = ["golden_full", "golden_no_eager", "golden_loss_only", "golden_fp32", "big_shapes_full", "golden_fwd_only",
            "golden_inline", "golden_no_aux"]


def child(v):
    import types
    import torch
    import torch.nn as nn
    from tests.golden_util import load
    import tests.test_llava_modules_gpu as TL
    from competesmoe_amd.graphs import GraphedStep
    dev = "cuda"
    if v.startswith("golden"):
        fx = load("llava_smoe_fp32" if v == "golden_fp32" else "llava_smoe_bf16")
        layer, dt = TL.build_layer(fx)
        x, dy = fx["x"].to(dev), fx["dy"].to(dev)
    else:
        from competesmoe_amd.moe import get_moe
        D, F, E, K, T = 256, 512, 8, 2, 512
        args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001)
        experts = nn.ModuleList([nn.Sequential(nn.Linear(D, F), nn.GELU(), nn.Linear(F, D)) for _ in range(E)])
        layer = get_moe("smoe")(D, D, E, K, experts, args).to(dev).bfloat16()
        x = torch.randn(2, T // 2, D, device=dev).bfloat16()
        dy = torch.randn_like(x)
    if v not in ("golden_no_eager",):
        xg = x.clone().requires_grad_(True)
        out, aux, _, _ = layer(xg)
        ((out.float() * dy.float()).sum() + aux.float()).backward()
        torch.cuda.synchronize()

    def fn(xs):
        out, aux, _, _ = layer(xs)
        if v == "golden_no_aux":
            return (out.float() * dy.float()).sum(), out
        loss = (out.float() * dy.float()).sum() + aux.float()
        return loss if v == "golden_loss_only" else (loss, out, aux)

    print("capture begin", flush=True)
    if v == "golden_fwd_only":
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            fn(x)
            fn(x)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s), torch.no_grad():
            r = fn(x)
        print("capture end", flush=True)
        g.replay()
    elif v == "golden_inline":
        xs = x.clone().requires_grad_(True)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                for p in layer.parameters():
                    p.grad = None
                xs.grad = None
                fn(xs)[0].backward()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for p in layer.parameters():
            p.grad = None
        xs.grad = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            r = fn(xs)
            r[0].backward()
        print("capture end", flush=True)
        g.replay()
    else:
        step = GraphedStep(fn, [x.clone().requires_grad_(True)], list(layer.parameters()))
        print("capture end", flush=True)
        step(x)
    torch.cuda.synchronize()
    print("replayed", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for pc in VARIANTS:
        r = subprocess.run([sys.executable, __file__, pc], capture_output=True, text=True, timeout=300)
        stage = [ln for ln in r.stdout.splitlines() if ln in ("capture begin", "capture end", "replayed")]
        err = [ln for ln in r.stderr.splitlines() if "Error" in ln or "error" in ln or "Fatal" in ln or "Warning" in ln][:3]
        print(f"{pc:18s} rc={r.returncode:4d} reached={stage[-1] if stage else '-':14s} {' | '.join(e[:160] for e in err)}", flush=True)
