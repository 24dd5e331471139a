#!/usr/bin/env python3
"""Which launch breaks hipGraph capture?  Captures ONE piece of the layer per child process (a crash inside the HIP runtime kills
only that child) and prints a table.  Usage: python tools/graph_bisect.py            (runs every piece)
                                              python tools/graph_bisect.py PIECE      (child mode)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PIECES = ["torch_only", "gate_logits", "router_select", "bin_tokens", "dispatch", "gemm_generic", "gemm_v1", "gemm_v2", "wgrad_v1",
          "wgrad_v2", "combine", "combine_bwd", "colsum", "router_aux", "layer_fwd", "layer_fwd_bwd"]


def child(piece):
    import torch
    import torch.nn as nn
    from competesmoe_amd import ops, _lib as L
    dev = "cuda"
    torch.manual_seed(0)
    T, D, F, E, K = 512, 256, 512, 8, 2
    x = torch.randn(T, D, device=dev).bfloat16()
    wg = torch.randn(E, D, device=dev).bfloat16()
    logits = ops.gate_logits(x, wg)
    sm, idx, w = ops.router_select(logits, K, L.SEL_SOFTMAX, True)
    bins = ops.bin_tokens(idx, E)
    xs = ops.dispatch_tokens(x, bins)
    W1 = [torch.randn(F, D, device=dev).bfloat16() / 16 for _ in range(E)]
    p1 = ops.ptr_array(W1, dev)
    big = piece in ("gemm_v2", "wgrad_v2")
    if big:
        T2 = 8192
        x2 = torch.randn(T2, 1024, device=dev).bfloat16()
        idx2 = torch.randint(0, E, (T2, 1), device=dev, dtype=torch.int32)
        bins2 = ops.bin_tokens(idx2, E)
        W2 = [torch.randn(1024, 1024, device=dev).bfloat16() / 32 for _ in range(E)]
        p2 = ops.ptr_array(W2, dev)
        gout = torch.empty(E, 1024, 1024, device=dev, dtype=torch.bfloat16)

    def run():
        if piece == "torch_only":
            return (x.float() * 2).sum()
        if piece == "gate_logits":
            return ops.gate_logits(x, wg)
        if piece == "router_select":
            return ops.router_select(logits, K, L.SEL_SOFTMAX, True)
        if piece == "bin_tokens":
            return ops.bin_tokens(idx, E).offsets
        if piece == "dispatch":
            return ops.dispatch_tokens(x, bins)
        if piece == "gemm_generic":
            return ops.grouped_gemm(xs, p1, L.B_NK, D, F, bins.offsets, E, force_generic=True)
        if piece == "gemm_v1":
            return ops.grouped_gemm(xs, p1, L.B_NK, D, F, bins.offsets, E)
        if piece == "gemm_v2":
            return ops.grouped_gemm(x2, p2, L.B_NK, 1024, 1024, bins2.offsets, E)
        if piece == "wgrad_v1":
            out = torch.empty(E, D, D, device=dev, dtype=torch.bfloat16)
            return ops.grouped_wgrad(xs, xs, bins.offsets, E, out, ops.ptr_table(out, E, D * D * 2))
        if piece == "wgrad_v2":
            return ops.grouped_wgrad(x2, x2, bins2.offsets, E, gout, ops.ptr_table(gout, E, 1024 * 1024 * 2), xcd_order=bins2.xcd_order)
        if piece == "combine":
            return ops.combine(xs, bins, idx, w, L.COMBINE_SEQ, T)
        if piece == "combine_bwd":
            return ops.combine_bwd(x, xs, bins, w)[0]
        if piece == "colsum":
            out = torch.empty(E, D, device=dev, dtype=torch.bfloat16)
            return ops.grouped_colsum(xs, bins.offsets, E, out, ops.ptr_table(out, E, D * 2))
        if piece == "router_aux":
            return ops.router_aux(logits.view(2, T // 2, E), sm.view(2, T // 2, E), idx.view(2, T // 2, K))[0]
        raise KeyError(piece)

    if piece.startswith("layer"):
        import types
        from competesmoe_amd.moe import get_moe
        args = types.SimpleNamespace(balance_loss_coef=0.01, router_z_loss_coef=0.001)
        experts = nn.ModuleList([nn.Sequential(nn.Linear(D, F), nn.GELU(), nn.Linear(F, D)) for _ in range(E)])
        layer = get_moe("smoe")(D, D, E, K, experts, args).to(dev).bfloat16()
        xl = x.view(2, T // 2, D).clone().requires_grad_(piece == "layer_fwd_bwd")

        def run():                                       # noqa: F811
            out, aux, _, _ = layer(xl)
            loss = out.float().sum() + aux.float()
            if piece == "layer_fwd_bwd":
                for p in layer.parameters():
                    p.grad = None
                xl.grad = None
                loss.backward()
            return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capture begin", flush=True)
    with torch.cuda.graph(g, stream=s):
        r = run()
    print("capture end", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for pc in PIECES:
        r = subprocess.run([sys.executable, __file__, pc], capture_output=True, text=True, timeout=300)
        stage = [ln for ln in r.stdout.splitlines() if ln in ("capture begin", "capture end", "replayed")]
        err = [ln for ln in r.stderr.splitlines() if "Error" in ln or "error" in ln or "Fatal" in ln][:2]
        print(f"{pc:16s} rc={r.returncode:4d} reached={stage[-1] if stage else '-':14s} {' | '.join(err)}", flush=True)
