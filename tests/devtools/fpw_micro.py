"""Per-layer time of the float (bf16) pointwise passes at B=256: stats / emit / reduce / dc, the data gradient and the weight gradient, each
launched alone on the layer's saved input (HIP events, 10 launches).  usage: python tests/devtools/fpw_micro.py [batch]"""
import sys, torch
sys.path.insert(0, ".")
import frostnet_amd.frostnet as F
from frostnet_amd import _lib as L
from frostnet_amd._lib import call, ptr, stream

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = F.frostnet_large_1_0().cuda().train()
x = torch.randn(B, 3, 224, 224, device="cuda")
r = m.hip_runner()
with torch.enable_grad():
    y = m(x)                      # recorded forward: every layer keeps its input
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = [0.0] * 6
print(f"{'layer':34s} {'npix':>9s} {'cin':>5s} {'cout':>5s} |  stats   emit    red     dc  dgrad  wgrad (us)   | GB/s stats emit red dc")
for l in r.layers:
    if l.kind != 0 or l.x is None: continue
    a = l.x
    npix, cin, cout = a.npix, a.c, l.cout
    gy = torch.randn(npix * cout + 64, device="cuda").to(torch.bfloat16).view(torch.int16)
    out = torch.empty(npix * cout + 64, dtype=torch.int16, device="cuda")
    dx = torch.empty(npix * cin + 64, dtype=torch.int16, device="cuda")
    gw = torch.zeros(cout * cin, device="cuda")
    f = r._fn["frost_float_pw"]
    t = [timed(lambda: call(f, l.desc_ptr, ptr(a.buf), ptr(l.pack), npix, cin, cout, int(l.relu), 0, None, 0, None, 0, stream())),
         timed(lambda: call(f, l.desc_ptr, ptr(a.buf), ptr(l.pack), npix, cin, cout, int(l.relu), 1, None, 0, ptr(out), cout, stream())),
         timed(lambda: call(f, l.desc_ptr, ptr(a.buf), ptr(l.pack), npix, cin, cout, int(l.relu), 2, ptr(gy), cout, None, 0, stream())),
         timed(lambda: call(f, l.desc_ptr, ptr(a.buf), ptr(l.pack), npix, cin, cout, int(l.relu), 3, ptr(gy), cout, ptr(out), cout, stream())),
         timed(lambda: call("frost_infer_pw", ptr(gy), ptr(l.pack_t), None, npix, cout, cin, 0, ptr(dx), stream())),
         timed(lambda: call(r._fn["frost_float_pw_wgrad"], ptr(gy), ptr(a.buf), npix, cin, cin, cout, ptr(gw), cin, stream()))]
    by = [2 * npix * cin, 2 * npix * (cin + cout), 2 * npix * (cin + cout), 2 * npix * (cin + 2 * cout)]
    for i in range(6): tot[i] += t[i]
    print(f"{l.name:34s} {npix:9d} {cin:5d} {cout:5d} | " + " ".join(f"{v:6.1f}" for v in t) + "   | " + " ".join(f"{b / v / 1e3:5.0f}" for b, v in zip(by, t)))
print("total ms: " + " ".join(f"{v / 1e3:.2f}" for v in tot))
