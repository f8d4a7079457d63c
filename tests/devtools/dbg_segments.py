"""Stress the eager-vs-graph-segment equivalence (tests/test_gpu_dp.py::test_graph_segments_replay_equals_eager) with per-parameter diagnostics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from frostnet_amd import frostnet as F
from frostnet_amd.optimizer import QSGD
from frostnet_amd.parallel import SegmentedStep
from test_gpu_dp import _shard

def run(graph, steps=2):
    torch.manual_seed(0)
    model = F.frostnet_quant_small_1_0(drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda().train()
    opt = QSGD([{"params": [p]} for p in model.parameters()], lr=1e-3, momentum=0.9, nesterov=True)
    seg = SegmentedStep(model.hip_runner(), torch.nn.CrossEntropyLoss(), nbuckets=4)
    x, tgt = _shard(0)
    x, tgt = x.cuda(), tgt.cuda()
    seg.run_eager(x, tgt)
    opt.step()
    hist = []
    if graph:
        seg.capture(x, tgt)
    for _ in range(steps):
        plan = opt.prepare_step()
        if graph:
            seg.replay()
        else:
            seg.run_eager(x, tgt)
        hist.append(model.hip_runner().grad_arena.clone())
        opt.launch(plan)
    torch.cuda.synchronize()
    names = [n for n, _ in model.named_parameters()]
    return names, [p.detach().clone() for p in model.parameters()], hist

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ref = None
for r in range(reps):
    for graph in (False, True):
        names, ps, hist = run(graph)
        if ref is None:
            ref = (ps, hist)
            continue
        rel_p = float(torch.cat([(a - b).flatten() for a, b in zip(ps, ref[0])]).norm() / torch.cat([b.flatten() for b in ref[0]]).norm())
        rel_g = [float((a - b).norm() / b.norm()) for a, b in zip(hist, ref[1])]
        flag = "BAD" if rel_p > 1e-5 else "ok"
        print(f"rep {r} graph={graph}: params rel {rel_p:.2e} grads rel {['%.2e' % v for v in rel_g]} {flag}", flush=True)
        if rel_p > 1e-5:
            worst = sorted(((float((a - b).norm() / (b.norm() + 1e-12)), n) for a, b, n in zip(ps, ref[0], names)), reverse=True)[:6]
            print("   worst params:", worst, flush=True)
            off = 0
            for n, p in zip(names, ps):
                k = p.numel()
                d = float((hist[0][off:off + k] - ref[1][0][off:off + k]).norm() / (ref[1][0][off:off + k].norm() + 1e-12))
                if d > 1e-2:
                    print(f"   step-0 grad {n}: rel {d:.2e}", flush=True)
                off += k
