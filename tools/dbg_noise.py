"""dev: run-to-run and batch-permutation differences of the parameter gradients of one QAT train step at full size."""
import copy, os, sys, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from frostnet_amd import frostnet as F
B, R = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(1882)
model = F.frostnet_quant_large_1_0(drop_rate=0.0)
F.qat_prepare(model, version=0)
model.cuda().train()
x = torch.randn(B, 3, R, R, device="cuda"); tgt = torch.randint(0, 1000, (B,), device="cuda")
for _ in range(2):
    torch.nn.functional.cross_entropy(model(x), tgt).backward()
state = copy.deepcopy(model.state_dict())
def run(xs, ts):
    model.load_state_dict(state); model.zero_grad(set_to_none=True)
    y = model(xs); torch.nn.functional.cross_entropy(y, ts).backward(); torch.cuda.synchronize()
    return y.detach().clone(), [(n, p.grad.detach().double().clone()) for n, p in model.named_parameters() if p.dim() == 4]
perm = torch.randperm(B, device="cuda")
a, b, c = run(x, tgt), run(x, tgt), run(x[perm], tgt[perm])
def show(tag, u, v):
    errs = [(float((p - q).norm() / q.norm()), n) for (n, p), (_, q) in zip(u[1], v[1])]
    print(tag, "first 6:", [f"{e:.1e}" for e, _ in errs[:6]], "last 4:", [f"{e:.1e}" for e, _ in errs[-4:]], "max", max(errs))
show("same-batch", a, b); show("permuted ", a, c)
print("logit equal same-batch:", bool(torch.equal(a[0], b[0])), " maxdiff permuted:", float((a[0][perm] - c[0]).abs().max()))
