"""conv1 -> conv2 pair of a bottleneck: fused (frost_block_expand_dw_stats) vs layer-by-layer, bit comparison + timing.
usage (GPU box): python tests/devtools/blk_pair.py "cin,cexp,H,k[,cout]" ... [--n 512]     (cout: append the reduce_conv, linear)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from frostnet_amd import _lib as L, engine

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(sys.argv[sys.argv.index("--n") + 1]) if "--n" in sys.argv else 512
dev = "cuda"


def build(cin, cexp, k, seed, cout=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    E, qa = engine.Engine(dev), engine.QArena(8, dev)
    def layer(name, kind, ci, co, kk, relu=True):
        w = (torch.randn(co, 1 if kind == "dw" else ci, kk, kk, generator=g) * (2.0 / (ci if kind == "pw" else kk * kk)) ** 0.5).to(dev).requires_grad_(True)
        gamma = (torch.rand(co, generator=g) * 0.5 + 0.75).to(dev).requires_grad_(True)
        beta = (torch.rand(co, generator=g) * 0.2 - 0.05).to(dev).requires_grad_(True)
        l = engine.ConvLayer(name, kind, w, gamma, beta, torch.zeros(co, device=dev), torch.ones(co, device=dev), torch.zeros((), dtype=torch.int64, device=dev), None, kk, 1, relu,
                             qa.alloc(), qa.alloc())
        return E.add_layer(l)
    l1, l2 = layer("c1", "pw", cin, cexp, 1), layer("c2", "dw", cexp, cexp, k)
    l3 = layer("c3", "pw", cexp, cout, 1, relu=False) if cout else None
    qx = qa.alloc(); qa.set_qparams(qx, 0.02, 3)
    return E, qa, l1, l2, l3, qx


for a in args:
    f = [int(v) for v in a.split(",")]
    cin, cexp, H, k = f[:4]; cout = f[4] if len(f) > 4 else 0
    res = []
    for fused in (0, 1):
        E, qa, l1, l2, l3, qx = build(cin, cexp, k, 7, cout)
        g = torch.Generator(device="cpu").manual_seed(99)
        x = E.new_act(N, H, H, cin, qx)
        x.buf[: x.numel] = torch.randint(-128, 128, (x.numel,), dtype=torch.int16, generator=g).to(torch.int8).to(dev)
        def fwd():
            E.begin_step()
            if fused:
                assert E.pair_fusable(l1, l2, x, True, True)
                y = E.conv_pair(l1, l2, x, l3=l3)
                assert l3 is None or y.kept_next is not None
            else:
                y = E.conv(l2, E.conv(l1, x))
            return E.conv(l3, y) if l3 is not None else y
        y2 = fwd()
        torch.cuda.synchronize()
        y1 = E.tape[0][3]; y3 = y2; y2 = E.tape[1][3]
        snap = dict(y1=y1.buf[: y1.numel].clone(), y2=y2.buf[: y2.numel].clone(), y3=y3.buf[: y3.numel].clone(), cint=(y3.cint[: y3.numel].clone() if l3 is not None and y3.cint is not None else None),
                    qy3=(l3.qy.clone() if l3 else None), coef3=(l3.coef.clone() if l3 else None), rv3=(l3.rvar.clone() if l3 else None), qy1=l1.qy.clone(), qy2=l2.qy.clone(), coef2=l2.coef.clone(), rm2=l2.rmean.clone(), rv2=l2.rvar.clone(),
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
