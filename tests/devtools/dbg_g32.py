import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge; ge.build()
from frostnet_amd import engine
from oracle import frost_oracle as O
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
for (cin, cout, H, N) in ((16, 96, 112, 2), (16, 96, 112, 1), (16, 96, 56, 4), (24, 144, 56, 4), (16, 96, 80, 2)):
    dev = "cuda"; seed = 5
    spec = O._convbn_spec("L", cin, cout, 1, 1)
    sd = O.synth_state([k_ for k_, _ in spec], [s_ for _, s_ in spec], seed)
    xi = np.clip(np.round(O.synth((N, cin, H, H), seed + 1) * 40 + 68), 0, 255).astype(np.uint8)
    E, qa = engine.Engine(dev), engine.QArena(4, dev)
    E.grad_fp32 = True; E._dbg = True
    w = sd["L.conv.0.weight"].to(dev).contiguous().requires_grad_(True)
    gamma, beta = sd["L.conv.1.weight"].to(dev).requires_grad_(True), sd["L.conv.1.bias"].to(dev).requires_grad_(True)
    l = engine.ConvLayer("L", "pw", w, gamma, beta, sd["L.conv.1.running_mean"].to(dev), sd["L.conv.1.running_var"].to(dev), torch.zeros((), dtype=torch.int64, device=dev), None, 1, 1, True, qa.alloc(), qa.alloc())
    E.add_layer(l)
    qx = qa.alloc(); qa.set_qparams(qx, 0.0231, 0)
    gr = T(O.synth((N, cout, H, H), seed + 2))
    E.begin_step()
    x = E.act_from_indices(T(xi), qx)
    y = E.conv(l, x, training=True, observe=True)
    y.grad = engine.float_to_grad(gr.to(dev), fp32=True)
    E.backward()
    torch.cuda.synchronize()
    dc = E._last_dc[: N * H * H * cout].view(-1, cout).double()
    xr = (x.buf[: x.numel].view(-1, cin).double() + 128.0) * 0.0231
    ref = dc.t() @ xr
    got = l.dwq.view(cout, cin).double()
    print((cin, cout, H, N), "dwq rel err", float((got - ref).norm() / ref.norm()), "npix", N * H * H)

print("---- S1 / S2 / dc invariants")
from frostnet_amd import _lib as L
for (cin, cout, H, N) in ((16, 96, 112, 2), (16, 96, 56, 4)):
    dev = "cuda"; seed = 5
    spec = O._convbn_spec("L", cin, cout, 1, 1)
    sd = O.synth_state([k_ for k_, _ in spec], [s_ for _, s_ in spec], seed)
    xi = np.clip(np.round(O.synth((N, cin, H, H), seed + 1) * 40 + 68), 0, 255).astype(np.uint8)
    E, qa = engine.Engine(dev), engine.QArena(4, dev)
    E.grad_fp32 = True; E._dbg = True
    w = sd["L.conv.0.weight"].to(dev).contiguous().requires_grad_(True)
    gamma, beta = sd["L.conv.1.weight"].to(dev).requires_grad_(True), sd["L.conv.1.bias"].to(dev).requires_grad_(True)
    l = engine.ConvLayer("L", "pw", w, gamma, beta, sd["L.conv.1.running_mean"].to(dev), sd["L.conv.1.running_var"].to(dev), torch.zeros((), dtype=torch.int64, device=dev), None, 1, 1, True, qa.alloc(), qa.alloc())
    E.add_layer(l)
    qx = qa.alloc(); qa.set_qparams(qx, 0.0231, 0)
    gr = T(O.synth((N, cout, H, H), seed + 2))
    E.begin_step()
    x = E.act_from_indices(T(xi), qx)
    y = E.conv(l, x, training=True, observe=True)
    y.grad = engine.float_to_grad(gr.to(dev), fp32=True)
    g_nhwc = y.grad[: y.numel].view(-1, cout).double().clone()
    E.backward()
    torch.cuda.synchronize()
    qw = l._g32_qw[: cout * cin].view(cout, cin).double()
    acc = (x.buf[: x.numel].view(-1, cin).double() + 128.0) @ qw.t()
    coef = l.coef.double()
    A, B, M, R, K1 = coef[0][:cout], coef[1][:cout], coef[2][:cout], coef[3][:cout], coef[4][:cout]
    qy = qa.get(l.qy)
    t = (A.float() * acc.float() + B.float()) / qy["scale"]
    mask = (t > 0) & (t <= 255.5)
    gy = g_nhwc * mask
    xhat = (acc - M) * R
    S1, S2 = gy.sum(0), (gy * xhat).sum(0)
    n = acc.shape[0]
    print((cin, cout, H, N), "S1 err", float((coef[5][:cout] - S1).norm() / S1.norm()), "S2 err", float((coef[6][:cout] - S2).norm() / S2.norm()),
          "mean xhat", float(xhat.mean(0).abs().max()), "mean xhat^2", float((xhat * xhat).mean(0).min()), float((xhat * xhat).mean(0).max()))
    dc_ref = K1 * (gy - S1 / n - xhat * S2 / n)
    dc = E._last_dc[: n * cout].view(-1, cout).double()
    print("   dc err", float((dc - dc_ref).norm() / dc_ref.norm()), "sum dc / |dc|", float(dc.sum(0).abs().max() / dc.abs().sum(0).max()))
    # the reference's own evaluation in fp32 vs fp64 for the same case
    for dt in (torch.float32, torch.float64):
        P, B_ = O.split_state({O.float_to_qat_key(k_): (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k_, v in sd.items()})
        qs = O.QState(B_)
        xo = ((T(xi).to(dt)) * 0.0231).requires_grad_(True)
        yo = O.convbn_qat(P, qs, "L", xo, 1, 0, 1, True, True)
        yo.backward(gr.to(dt))
        dWo = P["L.conv.0.weight"].grad.double().reshape(cout, cin)
        print("   oracle", dt, "dW vs device", float((l.w.grad.cpu().double().reshape(cout, cin) - dWo).norm() / dWo.norm()),
              "dgamma vs device", float((l.gamma.grad.cpu().double() - P["L.conv.0.bn.weight"].grad.double()).norm() / P["L.conv.0.bn.weight"].grad.double().norm()))
