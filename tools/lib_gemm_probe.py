"""What the vendor GEMM (hipBLASLt behind torch.matmul) reaches on this box at the headline layer's shapes -- a practical ceiling to read
the grouped kernels' TFLOP/s against (diagnostic; nothing in the product calls torch.matmul for these).
usage (GPU box): python tools/lib_gemm_probe.py"""
import torch

def timed(fn, iters=30, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

def main():
    import sys
    few = "--few" in sys.argv          # under rocprofv3 (tools/vendor_probe.sh): the two dense row-space shapes only, few launches
    dev = "cuda:0"
    torch.manual_seed(0)
    n, D, F, E = 65536, 4096, 11008, 64
    x = torch.randn(n, D, device=dev, dtype=torch.bfloat16)
    w1 = torch.randn(D, F, device=dev, dtype=torch.bfloat16) * 0.03
    h = torch.randn(n, F, device=dev, dtype=torch.bfloat16)
    w2 = torch.randn(F, D, device=dev, dtype=torch.bfloat16) * 0.03
    fl = 2.0 * n * D * F
    rows = []
    if few:
        for name, fn in (("dense NN  [n,D]x[D,F]", lambda: torch.matmul(x, w1)), ("dense NT  [n,F]x[D,F]^T", lambda: torch.matmul(h, w1.t())),
                         ("dense NN  [n,F]x[F,D]", lambda: torch.matmul(h, w2))):
            print(f"{name:52s} {timed(fn, iters=4, warm=2):8.3f} ms")
        return
    rows.append(("dense NN  [n,D]x[D,F]", timed(lambda: torch.matmul(x, w1))))
    rows.append(("dense NN  [n,F]x[F,D]", timed(lambda: torch.matmul(h, w2))))
    rows.append(("dense NT  [n,F]x[D,F]^T", timed(lambda: torch.matmul(h, w1.t()))))
    rows.append(("dense TN  [n,D]^T x [n,F]", timed(lambda: torch.matmul(x.t(), h)), 2.0 * n * D * F))
    xb = x.view(E, n // E, D)
    w1b = (torch.randn(E, D, F, device=dev, dtype=torch.bfloat16) * 0.03)
    hb = h.view(E, n // E, F)
    w2b = (torch.randn(E, F, D, device=dev, dtype=torch.bfloat16) * 0.03)
    rows.append(("bmm 64 x [1024,D]x[D,F]", timed(lambda: torch.bmm(xb, w1b))))
    rows.append(("bmm 64 x [1024,F]x[F,D]", timed(lambda: torch.bmm(hb, w2b))))
    rows.append(("bmm 64 x [1024,D]^T x [1024,F] (wgrad)", timed(lambda: torch.bmm(xb.transpose(1, 2), hb))))
    sq = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    rows.append(("dense NN 8192^3", timed(lambda: torch.matmul(sq, sq)), 2.0 * 8192 ** 3))
    for r in rows:
        f = r[2] if len(r) > 2 else fl
        print(f"{r[0]:52s} {r[1]:8.3f} ms  {f / r[1] / 1e9:8.1f} TFLOP/s")

if __name__ == "__main__":
    main()
