"""Per-pass times of single ConvBN(ReLU) layers at production size (HIP events around every tagged launch, eager).
Usage (GPU box): python tests/devtools/pw_micro.py "cin,cout,H[,k,stride,relu]" ... [--n 512] [--iters 20]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from frostnet_amd import _lib as L, engine

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 512
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
dev = "cuda"
torch.manual_seed(0)
for a in args:
    f = [int(v) for v in a.split(",")]
    cin, cout, H = f[:3]
    k = f[3] if len(f) > 3 else 1
    s = f[4] if len(f) > 4 else 1
    relu = f[5] if len(f) > 5 else 1
    kind = "dw" if k > 1 else "pw"
    E, qa = engine.Engine(dev), engine.QArena(4, dev)
    w = (torch.randn(cout, 1 if kind == "dw" else cin, k, k, device=dev) * 0.1).requires_grad_(True)
    gamma, beta = (torch.rand(cout, device=dev) + 0.5).requires_grad_(True), (torch.randn(cout, device=dev) * 0.1).requires_grad_(True)
    l = engine.ConvLayer("L", kind, w, gamma, beta, torch.zeros(cout, device=dev), torch.ones(cout, device=dev),
                         torch.zeros((), dtype=torch.int64, device=dev), None, k, s, bool(relu), qa.alloc(), qa.alloc())
    E.add_layer(l)
    qx = qa.alloc()
    qa.set_qparams(qx, 0.0231, 0 if relu else 117)
    xi = torch.randint(0, 200, (N, cin, H, H), dtype=torch.uint8)
    Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
    g = torch.randn(N, cout, Ho, Ho, device=dev)

    def step():
        E.begin_step()
        x = E.act_from_indices(xi, qx)
        x.needs_grad = True
        y = E.conv(l, x, training=True, observe=True)
        y.grad = engine.float_to_grad(g)
        E.backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    L.PROFILER = L.Profiler()
    for _ in range(iters):
        step()
    summ = L.PROFILER.summary()
    L.PROFILER = None
    tot = sum(v["total_ms"] for v in summ.values()) / iters * 1000
    print(f"{a:24s} npix={N * H * H:8d} " + " ".join(f"{k_.split('_', 1)[1]}={v['avg_ms'] * 1000:.1f}" for k_, v in summ.items() if k_ != "weight_prep") + f" | {tot:.0f} us")
