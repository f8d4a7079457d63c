"""dev: does a tensor written by one kernel get read back from the 256 MB memory-side cache by the next kernel?"""
import torch, time
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
big = torch.empty(3 * 1024**3 // 2, dtype=torch.int16, device=dev)          # 3 GB scratch to flush caches
for mb in (32, 64, 96, 128, 192, 256, 512, 1024):
    n = mb * 1024 * 1024 // 2
    src = torch.randn(n, device=dev).to(torch.bfloat16); dst = torch.empty_like(src)
    def wr(): dst.copy_(src)
    def rd(): return dst.float().sum() if False else torch.sum(dst, dtype=torch.float32)
    # A: write then read immediately (producer-consumer)   B: write, flush with a 3 GB fill, read
    def a(): wr(); rd()
    def b(): wr(); big.fill_(1); rd()
    def w_only(): wr()
    def wf(): wr(); big.fill_(1)
    ta, tb, tw, twf = t(a), t(b), t(w_only), t(wf)
    print(f"{mb:5d} MB: read after write {mb/1024/((ta-tw)*1e-3):7.0f} GB/s   read after flush {mb/1024/((tb-twf)*1e-3):7.0f} GB/s   (copy {2*mb/1024/(tw*1e-3):6.0f} GB/s)")
