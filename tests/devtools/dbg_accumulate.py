"""Run-to-run spread of the float model's gradients vs the accumulate-vs-sum error of
tests/test_gpu_surface.py::test_backward_accumulates_until_zero_grad, for a few (precision, batch, image) settings."""
import sys, torch
sys.path.insert(0, ".")
import frostnet_amd.frostnet as F

def run(prec, B, S):
    def make():
        torch.manual_seed(3)
        m = F.frostnet_small_1_0(drop_rate=0.0)
        m.float_precision = prec
        return m.cuda().train()
    x1, x2 = torch.randn(B, 3, S, S, device="cuda"), torch.randn(B, 3, S, S, device="cuda")
    t = torch.arange(B, device="cuda") % 1000
    ce = torch.nn.functional.cross_entropy
    flat = lambda m: torch.cat([p.grad.detach().reshape(-1) for p in m.parameters()]).clone()
    a = make(); ce(a(x1), t).backward(); g1 = flat(a); a.zero_grad(); ce(a(x2), t).backward(); g2 = flat(a)
    a3 = make(); ce(a3(x1), t).backward(); g1b = flat(a3)
    b = make(); ce(b(x1), t).backward(); ce(b(x2), t).backward(); gb = flat(b)
    rel = lambda u, v: float((u - v).norm() / v.norm())
    print(f"{prec} B={B} S={S}: acc-vs-sum {rel(gb, g1 + g2):.2e}   identical-run spread {rel(g1, g1b):.2e}   acc-vs-last {rel(gb, g2):.2e}", flush=True)

for cfg in (("bf16", 4, 64), ("bf16", 16, 128), ("bf16", 32, 160), ("fp32", 4, 64), ("fp32", 8, 96), ("fp32", 16, 128)):
    run(*cfg)
