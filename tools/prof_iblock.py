"""Dev: one bottleneck geometry of the fused bf16 inference kernel in a loop (for rocprofv3 counter passes).  usage: prof_iblock.py cin r cexp cout k stride H [batch] [th tw]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import _lib as L, infer as I
a = [int(v) for v in sys.argv[1:]]
cin, r, cexp, cout, k, s, H = a[:7]
n = a[7] if len(a) > 7 else 256
lib = L.load_library()
tile = (a[8], a[9]) if len(a) > 9 else I.pick_tile(lib, H, H, cin, r, cexp, cout, k, s)
print("tile", tile, "lds", lib.frost_infer_block_ok(H, H, cin, r, cexp, cout, k, s, tile[0], tile[1]))
dev = "cuda"
def pack(co, ci): return torch.randn(((co + 15) // 16) * ((ci + 31) // 32) * 64 * 8, device=dev).mul(0.05).to(torch.bfloat16).view(torch.int16)
x = torch.randn(n * H * H * cin + 64, device=dev).to(torch.bfloat16).view(torch.int16)
wsq, bsq = (pack(r, cin), torch.zeros(r + 16, device=dev)) if r else (None, None)
w1, b1 = (pack(cexp, r + cin), torch.zeros(cexp + 16, device=dev)) if cexp != cin or r else (None, None)
wdw, bdw = torch.randn(k * k * ((cexp + 15) // 16 * 16), device=dev) * 0.1, torch.zeros(cexp + 16, device=dev)
w3, b3 = pack(cout, cexp), torch.zeros(cout + 16, device=dev)
Ho = (H + 2 * ((k - 1) // 2) - k) // s + 1
y = torch.empty(n * Ho * Ho * cout + 64, dtype=torch.int16, device=dev)
def run():
    L.call("frost_infer_block", L.ptr(x), L.ptr(wsq), L.ptr(bsq), L.ptr(w1), L.ptr(b1), L.ptr(wdw), L.ptr(bdw), L.ptr(w3), L.ptr(b3), n, H, H, cin, r, cexp, cout, k, s,
           1 if (s == 1 and cin == cout) else 0, tile[0], tile[1], 0, 0, L.ptr(y), L.stream())
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("us per launch", e0.elapsed_time(e1) * 100)
