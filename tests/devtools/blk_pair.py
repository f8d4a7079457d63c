"""conv1 -> conv2 pair of a bottleneck: fused (frost_block_expand_dw_stats) vs layer-by-layer, bit comparison + timing.
usage (GPU box): python tests/devtools/blk_pair.py "cin,cexp,H,k" ... [--n 512]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from frostnet_amd import _lib as L, engine

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 512
dev = "cuda"


def build(cin, cexp, k, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    E, qa = engine.Engine(dev), engine.QArena(8, dev)
    def layer(name, kind, ci, co, kk):
        w = (torch.randn(co, 1 if kind == "dw" else ci, kk, kk, generator=g) * (2.0 / (ci if kind == "pw" else kk * kk)) ** 0.5).to(dev).requires_grad_(True)
        gamma = (torch.rand(co, generator=g) * 0.5 + 0.75).to(dev).requires_grad_(True)
        beta = (torch.rand(co, generator=g) * 0.2 - 0.05).to(dev).requires_grad_(True)
        l = engine.ConvLayer(name, kind, w, gamma, beta, torch.zeros(co, device=dev), torch.ones(co, device=dev), torch.zeros((), dtype=torch.int64, device=dev), None, kk, 1, True,
                             qa.alloc(), qa.alloc())
        return E.add_layer(l)
    l1, l2 = layer("c1", "pw", cin, cexp, 1), layer("c2", "dw", cexp, cexp, k)
    qx = qa.alloc(); qa.set_qparams(qx, 0.02, 3)
    return E, qa, l1, l2, qx


for a in args:
    cin, cexp, H, k = [int(v) for v in a.split(",")]
    res = []
    for fused in (0, 1):
        E, qa, l1, l2, qx = build(cin, cexp, k, 7)
        g = torch.Generator(device="cpu").manual_seed(99)
        x = E.new_act(N, H, H, cin, qx)
        x.buf[: x.numel] = torch.randint(-128, 128, (x.numel,), dtype=torch.int16, generator=g).to(torch.int8).to(dev)
        def fwd():
            E.begin_step()
            if fused:
                assert E.pair_fusable(l1, l2, x, True, True)
                return E.conv_pair(l1, l2, x)
            return E.conv(l2, E.conv(l1, x))
        y2 = fwd()
        torch.cuda.synchronize()
        y1 = E.tape[-2][3]
        snap = dict(y1=y1.buf[: y1.numel].clone(), y2=y2.buf[: y2.numel].clone(), qy1=l1.qy.clone(), qy2=l2.qy.clone(), coef2=l2.coef.clone(), rm2=l2.rmean.clone(), rv2=l2.rvar.clone(),
                    stats=None)
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fwd()
        e1.record(); torch.cuda.synchronize()
        L.PROFILER = L.Profiler()
        for _ in range(5):
            fwd()
        summ = L.PROFILER.summary(); L.PROFILER = None
        print("   ", "fused" if fused else "layer", " ".join(f"{k2}={v['avg_ms'] * 1e3:.1f}" for k2, v in summ.items()), flush=True)
        res.append((snap, e0.elapsed_time(e1) / 20 * 1e3))
    (a0, t0), (a1, t1) = res
    bad = [k2 for k2 in a0 if a0[k2] is not None and not torch.equal(a0[k2].view(torch.uint8), a1[k2].view(torch.uint8))]
    print(f"{a:<18} layerwise {t0:7.1f} us   fused {t1:7.1f} us   mismatching: {bad if bad else 'none'}", flush=True)
    if bad:
        for k2 in bad:
            d = (a0[k2].float() - a1[k2].float()).abs()
            print("   ", k2, "max", float(d.max()), "frac", float((d > 0).float().mean()))
