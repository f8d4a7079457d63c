"""Library GEMM times (hipBLASLt via torch) at the low-resolution pointwise shapes -- a yardstick for the hand-written kernels, not a code path."""
import torch, sys
shapes = [(100352, 104, 312), (100352, 312, 80), (100352, 104, 624), (100352, 624, 96), (100352, 144, 864), (25088, 864, 192),
          (25088, 240, 1440), (25088, 1440, 192), (25088, 288, 1728), (25088, 1728, 320), (25088, 320, 1280)]
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for M, K, N in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    tb = t(lambda: torch.mm(a, b))
    g = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    tw = t(lambda: torch.mm(g.t(), a))          # wgrad shape: [N x M] x [M x K]
    td = t(lambda: torch.mm(g, b.t()))          # dgrad shape
    ti = float("nan")
    try:
        Kp = (K + 15) // 16 * 16
        ai = torch.randint(-128, 127, (M, Kp), device="cuda", dtype=torch.int8); bi = torch.randint(-128, 127, (Kp, N), device="cuda", dtype=torch.int8)
        ti = t(lambda: torch._int_mm(ai, bi))
    except Exception as e:
        ti = float("nan")
    print(f"M={M:7d} K={K:5d} N={N:5d}  fwd bf16 {tb:7.1f} us  int8 {ti:7.1f} us   dgrad bf16 {td:7.1f}   wgrad bf16 {tw:7.1f}   (GMAC {M*K*N/1e9:.1f})")
