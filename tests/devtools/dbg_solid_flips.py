"""Where do device indices differ from BOTH the fp32 and the fp64 reference?  Per-channel counts and the channels' conditioning (|batch mean| / batch std
of the conv output: the device's single fma y = A*acc + B cancels a*mean against a*acc; the reference subtracts the mean first)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_prod as TP
from oracle import frost_oracle as O
from frostnet_amd import engine
name = sys.argv[1]
case = [c for c in TP.ALL_LAYERS if c[0] == name][0]
name, cin, cout, k, s, groups, H, N, relu, in_zp = case
seed = 9100 + 7 * TP.ALL_LAYERS.index(case)
dev = "cuda"; torch.set_num_threads(16)
sd = TP._layer_state(cin, cout, k, groups, seed)
in_scale = 0.0231
xi = np.clip(np.round(O.synth((N, cin, H, H), seed + 1) * 40 + 128 + (0 if in_zp else -60)), 0, 255).astype(np.uint8)
T = TP.T
P64, B64 = O.split_state({O.float_to_qat_key(k_): (v.clone().double() if v.is_floating_point() else v.clone()) for k_, v in sd.items()})
qs64 = O.QState(B64)
pre = {}
orig = qs64.fq_site
def spy(prefix, x, kind, observe=True):
    if prefix.endswith("conv.0.activation_post_process"): pre["y"] = x.detach().clone()
    return orig(prefix, x, kind, observe)
qs64.fq_site = spy
xo64 = ((T(xi.astype(np.float64)) - in_zp) * in_scale)
yo64 = O.convbn_qat(P64, qs64, "L", xo64, s, (k - 1) // 2, groups, bool(relu), True)
a = "L.conv.0.activation_post_process"
sc, zp = float(qs64.sd[a + ".scale"][0]), int(qs64.sd[a + ".zero_point"][0])
idx64 = O.fq_index(yo64, qs64.sd[a + ".scale"][0], qs64.sd[a + ".zero_point"][0])
t64 = pre["y"] / sc + zp
E, qa = engine.Engine(dev), engine.QArena(4, dev)
w = sd["L.conv.0.weight"].to(dev).contiguous().requires_grad_(True)
gamma, beta = sd["L.conv.1.weight"].to(dev).requires_grad_(True), sd["L.conv.1.bias"].to(dev).requires_grad_(True)
kind = "dw" if groups > 1 else "pw"
l = engine.ConvLayer("L", kind, w, gamma, beta, sd["L.conv.1.running_mean"].to(dev), sd["L.conv.1.running_var"].to(dev), torch.zeros((), dtype=torch.int64, device=dev), None, k, s, bool(relu), qa.alloc(), qa.alloc())
E.add_layer(l); qx = qa.alloc(); qa.set_qparams(qx, in_scale, in_zp)
E.begin_step(); x = E.act_from_indices(T(xi), qx); y = E.conv(l, x, training=True, observe=True)
yidx = y.indices().cpu(); torch.cuda.synchronize()
d = (yidx.to(torch.int16) - idx64.to(torch.int16))
bad = d != 0
print(f"{name}: device vs fp64 mismatches {int(bad.sum())} of {bad.numel()} ({float(bad.float().mean()):.2e}); signs +{int((d > 0).sum())} / -{int((d < 0).sum())}")
frac = (t64 - torch.floor(t64))[bad]
print("distance of the fp64 pre-rounding value from the nearest rounding boundary (steps) at the mismatches: median %.2e max %.2e" % (float((frac - 0.5).abs().median()), float((frac - 0.5).abs().max())))
perch = bad.sum((0, 2, 3))
c0 = pre["y"]  # post-BN pre-FQ y; conditioning of the conv output from the BN batch stats: use oracle internals
base = "L.conv.0"
wq = None
top = torch.argsort(perch, descending=True)[:8]
rs = torch.sqrt(B64[base + ".bn.running_var"] + 1e-5)
sf = P64[base + ".bn.weight"] / rs
with torch.no_grad():
    wfq = qs64.fq_site(base + ".weight_fake_quant", P64[base + ".weight"] * sf.reshape(-1, 1, 1, 1), qs64.wgt, observe=False)
    c = torch.nn.functional.conv2d(xo64, wfq, None, s, (k - 1) // 2, 1, groups) / sf.reshape(1, -1, 1, 1)
mu, sg = c.mean((0, 2, 3)), c.std((0, 2, 3))
for ch in top.tolist():
    print(f"  channel {ch}: {int(perch[ch])} mismatches; gamma {float(P64[base + '.bn.weight'][ch]):+.4f} sf {float(sf[ch]):+.4e} |mean|/std of c0 {float(mu[ch].abs() / sg[ch]):.2f}  A*acc scale: y range {float(pre['y'][:, ch].min()):+.3f}..{float(pre['y'][:, ch].max()):+.3f} (step {sc:.4f})")
print("median |mean|/std over channels %.2f; median mismatches per channel %.1f" % (float((mu.abs() / sg).median()), float(perch.float().median())))
# ---- are the device's quantised weights the reference's?
if kind == "pw":
    CT, KS = l.cpad // 16, l.kpad // 64
    pk = l.wq_pack[: CT * KS * 1024].cpu().view(CT, KS, 64, 16).numpy().astype(np.int32)
    dev_w = np.zeros((l.cpad, l.kpad), np.int32)
    for ln in range(64):
        dev_w[(np.arange(CT) * 16 + (ln & 15))[:, None, None], (np.arange(KS) * 64 + (ln >> 4) * 16)[None, :, None] + np.arange(16)[None, None, :]] = pk[:, :, ln, :]
    dev_w = dev_w[:cout, :cin]
    ws = float(qs64.sd[base + ".weight_fake_quant.scale"][0])
    for tag, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
        wr, g_, rv = sd["L.conv.0.weight"].to(dt), sd["L.conv.1.weight"].to(dt), sd["L.conv.1.running_var"].to(dt)
        sfr = g_ / torch.sqrt(rv + 1e-5)
        wsr = (wr * sfr.reshape(-1, 1, 1, 1)).reshape(cout, cin)
        inv = (torch.tensor(1.0, dtype=dt) / torch.tensor(ws if dt == torch.float64 else np.float32(ws), dtype=dt))
        q = torch.clamp(torch.round(wsr * inv), -128, 127).numpy().astype(np.int32)
        diff = np.argwhere(q != dev_w)
        print(f"quantised weights, device vs {tag} reference: {len(diff)} differ", [(int(a_), int(b_), int(dev_w[a_, b_]), int(q[a_, b_]), float(wsr[a_, b_] * inv)) for a_, b_ in diff[:6]])
    # ---- bit-level: device weight scale vs the fp32 reference's; numpy fp32 emulation of the device's formula for the differing weight
    P32, B32 = O.split_state({O.float_to_qat_key(k_): v.clone() for k_, v in sd.items()})
    qs32 = O.QState(B32)
    with torch.no_grad():
        O.convbn_qat(P32, qs32, "L", xo64.float(), s, (k - 1) // 2, groups, bool(relu), True)
    s_ref = qs32.sd[base + ".weight_fake_quant.scale"][0].numpy()
    s_dev = np.float32(qa.get(l.qw)["scale"])
    print("weight scale bits: device %08x reference %08x" % (s_dev.view(np.uint32), np.float32(s_ref).view(np.uint32)), "min/max dev", qa.get(l.qw)["min_val"], qa.get(l.qw)["max_val"],
          "ref", float(qs32.sd[base + ".weight_fake_quant.activation_post_process.min_val"]), float(qs32.sd[base + ".weight_fake_quant.activation_post_process.max_val"]))
    g32, rv32, w32 = sd["L.conv.1.weight"].numpy().astype(np.float32), sd["L.conv.1.running_var"].numpy().astype(np.float32), sd["L.conv.0.weight"].numpy().astype(np.float32).reshape(cout, cin)
    for a_, b_ in diff[:3]:
        sf_ = np.float32(g32[a_] / np.sqrt(np.float32(rv32[a_] + np.float32(1e-5))))
        x_ = np.float32(w32[a_, b_] * sf_)
        for tag2, sc_ in (("dev scale", s_dev), ("ref scale", np.float32(s_ref))):
            inv_ = np.float32(np.float32(1.0) / sc_)
            print(f"   weight ({a_},{b_}): w*sf = {x_!r} ({x_.view(np.uint32):08x}); with {tag2}: x*inv = {np.float32(x_ * inv_)!r}")
