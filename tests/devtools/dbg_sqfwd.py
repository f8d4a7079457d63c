"""dev: frost_sq_fwd vs the two launches it replaces, which tensors / coefficient rows differ (python tests/devtools/dbg_sqfwd.py [cin r H n])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import engine, _lib as L
cin, r, H, n = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (80, 24, 14, 7)
dev = "cuda"


def run(persist):
    engine._SQ_PERSIST = persist
    g = torch.Generator(device="cpu").manual_seed(77)
    E, qa = engine.Engine(dev), engine.QArena(8, dev)
    w = (torch.randn(r, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).to(dev).requires_grad_(True)
    gamma = (torch.rand(r, generator=g) * 0.5 + 0.75).to(dev).requires_grad_(True)
    beta = (torch.rand(r, generator=g) * 0.2 - 0.05).to(dev).requires_grad_(True)
    l = E.add_layer(engine.ConvLayer("sq", "pw", w, gamma, beta, torch.zeros(r, device=dev), torch.ones(r, device=dev), torch.zeros((), dtype=torch.int64, device=dev),
                                     None, 1, 1, True, qa.alloc(), qa.alloc()))
    qx, qcat = qa.alloc(), qa.alloc()
    qa.set_qparams(qx, 0.021, 117)
    qx[4], qx[5] = -2.4, 2.9
    outs = []
    for step in range(2):
        x = E.new_act(n, H, H, cin, qx)
        x.buf[: x.numel] = torch.randint(-128, 128, (x.numel,), dtype=torch.int16, generator=g).to(torch.int8).to(dev)
        E.begin_step()
        sq = E.conv(l, x, True, True, cat=(x.q, qcat))
        y = E.cat(sq, x, qcat, True)
        torch.cuda.synchronize()
        outs.append(dict(sq=sq.buf[: sq.numel].clone(), cat=y.buf[: y.numel].clone(), qcat=qcat.clone(), qy=l.qy.clone(), coef=l.coef.clone(), rmean=l.rmean.clone(), rvar=l.rvar.clone(),
                         nbt=l.nbt.clone(), ctl=l.fin_counter.clone(), stats=l.stats.clone() if hasattr(l, "stats") else None))
    return outs


a, b = run(True), run(False)
for step, (sa, sb) in enumerate(zip(a, b)):
    for k in sa:
        if sa[k] is None or k == "ctl":
            continue
        eq = torch.equal(sa[k], sb[k])
        print(step, k, "equal" if eq else "DIFFERENT")
        if not eq and k == "coef":
            for row in range(sa[k].shape[0]):
                d = (sa[k][row] != sb[k][row]).nonzero().flatten().tolist()
                if d:
                    print("   row", row, "cols", d[:12], "persist", sa[k][row][d[:4]].tolist(), "layer", sb[k][row][d[:4]].tolist())
    print("   ctl[32:40]", sa["ctl"][32:40].tolist())
