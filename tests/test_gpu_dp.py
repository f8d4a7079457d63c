"""Data-parallel path on the device (SURVEY 8e 'parity check without a cluster'): two ranks (both on cuda:0, gloo transport -- the
test box has one GPU; on the 8-GPU node the same code runs one rank per GPU over RCCL) each run the HIP forward + backward on
their shard through frostnet_amd.parallel.SegmentedStep; after the bucketed all-reduce every rank's gradient arena must equal the
MEAN of the per-shard gradients -- checked against the ranks' own pre-reduction gradients (exact) and against the CPU oracle run on
each shard from identical weights (bf16-storage tolerance).  Also: the hipGraph-segment replay of the step equals the eager step,
and `bench.py --gpus 2` really runs two ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

pytestmark = pytest.mark.gpu
TAIL_TOL = 2e-2          # classifier bias gradient of the data-parallel step vs the oracle (element-wise, relative L2)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODE, RES, B = "small", 64, 2


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _shard(rank):
    return T(O.synth((B, 3, RES, RES), 40 + rank)), torch.tensor([3 + rank, 500 + rank])


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from frostnet_amd import frostnet as F
    from frostnet_amd.parallel import SegmentedStep, broadcast_model
    cfg = O.net_cfg(MODE, 1.0)
    spec = O.float_state_spec(cfg)
    model = F.MODEL_REGISTRY[f"frostnet_quant_{MODE}_1_0"](drop_rate=0.0)
    model.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], 5000 + 17 * rank))    # ranks start DIFFERENT ...
    F.qat_prepare(model, version=0)
    model.cuda().train()
    broadcast_model(model)                                                                               # ... rank 0's weights win
    runner = model.hip_runner()
    seg = SegmentedStep(runner, torch.nn.CrossEntropyLoss(), nbuckets=3)
    x, tgt = _shard(rank)
    # this rank's own gradient = the bucket as it stands when the backward hands it to the exchange (same run: a second backward would
    # differ by the order of its fp32 atomics and the bf16 roundings that order flips, 1e-7 ... 3e-3, see the segments test below)
    local = torch.zeros_like(runner.grad_arena)
    exchange = seg._reduce

    def snapshot_then_reduce(i):
        for lo, hi in seg.cuts[i][1]:
            local[lo:hi].copy_(runner.grad_arena[lo:hi])
        exchange(i)
    seg._reduce = snapshot_then_reduce
    loss = seg.run_eager(x.cuda(), tgt.cuda())
    seg.finish()
    torch.cuda.synchronize()
    q.put((rank, runner.grad_arena.cpu().numpy().copy(), local.cpu().numpy().copy(), float(loss), len(seg.cuts),
           float(model.state_dict()["conv1.conv.0.weight"].abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_mean_of_shard_gradients():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    assert res[0][4] >= 2                                          # really bucketed
    assert res[0][5] == res[1][5]                                  # broadcast made the replicas identical
    total_local = res[0][2] + res[1][2]        # each rank's loss was pre-divided by the world size: the SUM of the local arenas is the mean gradient
    for r in range(2):                         # all-reduced arena == that sum, bit for bit (one fp32 addition per element, commutative)
        assert np.array_equal(res[r][1], total_local)
    assert float(np.linalg.norm(res[0][2] - res[1][2])) > 0.1 * float(np.linalg.norm(total_local))     # the shards really differ
    assert np.array_equal(res[0][1], res[1][1])
    # oracle: each shard from rank 0's weights, gradients averaged
    torch.set_num_threads(16)
    cfg = O.net_cfg(MODE, 1.0)
    spec = O.float_state_spec(cfg)
    grads = []
    for r in range(2):
        P, Bf = O.make_state(spec, 5000, True)
        qs = O.QState(Bf)
        x, tgt = _shard(r)
        torch.nn.functional.cross_entropy(O.frostnet_forward(P, qs, cfg, x, True, True), tgt).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in P.values()]).numpy())
    expect = (grads[0] + grads[1]) / 2
    got = res[0][1]
    # A train-mode QAT network amplifies single index flips chaotically (SURVEY H-2: the reference moves 5.5e-2 against itself when only
    # its thread count changes), so end to end the device gradient is only sanity-bounded against the oracle here; the exact statement
    # "post-all-reduce arena == mean over shards of the oracle's gradients" is tests/test_cpu_surface.py::
    # test_data_parallel_shard_oracle_mean_gradient_gloo_world2, and the per-layer gradient parity is tests/test_gpu_prod.py.
    off, ratios = 0, {}
    for n_, p in P.items():
        ratios[n_] = np.linalg.norm(got[off:off + p.numel()]) / (np.linalg.norm(expect[off:off + p.numel()]) + 1e-30)
        off += p.numel()
    med = float(np.median(list(ratios.values())))
    # the TAIL of the backward.  Element-wise the classifier / last_layer WEIGHT gradients are already 0.24 - 0.30 from the oracle's here (measured:
    # isolated index flips of the train-mode forward move the logits, hence dlogits, at this batch size), so the element-wise bound is put on the quantity that
    # averages them out -- the classifier BIAS gradient (sum of dlogits over the batch: 4.5e-3 measured) -- and the weight tensors of the tail get a NORM band that a
    # 2x scaling bug or a wrong 1 / world factor cannot pass; the loose band is left to the deep layers
    off, tail, tnorm = 0, {}, {}
    for n_, p in P.items():
        if n_.startswith("classifier.") or n_.startswith("last_layer."):
            a_, b_ = got[off:off + p.numel()], expect[off:off + p.numel()]
            tail[n_] = float(np.linalg.norm(a_ - b_) / (np.linalg.norm(b_) + 1e-30))
            tnorm[n_] = float(np.linalg.norm(a_) / (np.linalg.norm(b_) + 1e-30))
        off += p.numel()
    print(f"[dp] gradient-norm ratio device/oracle of the all-reduced arena: median {med:.3f}, classifier {ratios['classifier.2.weight']:.3f}; "
          f"tail gradients vs the oracle's shard mean, element-wise: { {k: round(v, 4) for k, v in tail.items()} }, norm ratios: { {k: round(v, 3) for k, v in tnorm.items()} }")
    assert 0.33 < med < 3.0
    assert tail["classifier.2.bias"] <= TAIL_TOL, tail
    assert all(0.8 <= v <= 1.25 for v in tnorm.values()), tnorm


def test_graph_segments_replay_equals_eager():
    """SegmentedStep.capture/replay (what bench.py runs for N>1) against the eager step, one process, no process group.

    Both sides start from the SAME restored state and take ONE step: the forward is bit-reproducible (integer / fp64 statistics), the
    backward differs by the order of fp32 atomics and the stochastic bf16 roundings that order flips.  A multi-step trajectory is not
    compared: at B = 2 (2x2 final maps) a 1e-9 parameter difference moves a zero-point or an index in the next forward and the runs
    branch (tests/devtools/dbg_segments.py: eager vs eager branches just as often as eager vs graph)."""
    sys.path.insert(0, ROOT)
    from frostnet_amd import frostnet as F
    from frostnet_amd.optimizer import QSGD
    from frostnet_amd.parallel import SegmentedStep

    torch.manual_seed(0)
    model = F.frostnet_quant_small_1_0(drop_rate=0.0)
    F.qat_prepare(model, version=0)
    model.cuda().train()
    opt = QSGD([{"params": [p]} for p in model.parameters()], lr=1e-3, momentum=0.9, nesterov=True)     # is_warmup: no noise
    runner = model.hip_runner()
    seg = SegmentedStep(runner, torch.nn.CrossEntropyLoss(), nbuckets=4)
    x, tgt = _shard(0)
    x, tgt = x.cuda(), tgt.cuda()
    seg.run_eager(x, tgt)                 # descriptor tables and observer state exist before the capture
    opt.step()
    torch.cuda.synchronize()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    seg.capture(x, tgt)
    assert len(seg.graphs) == len(seg.cuts) >= 2

    def one(graph):
        model.load_state_dict(sd0)
        loss = seg.replay() if graph else seg.run_eager(x, tgt)
        torch.cuda.synchronize()
        post = torch.cat([v.detach().float().reshape(-1) for k, v in model.state_dict().items() if "running_" in k or "min_val" in k or "max_val" in k])
        return float(loss), runner.grad_arena.clone(), torch.nan_to_num(post, posinf=0.0, neginf=0.0)     # never-used sites keep +-inf
    le, ge_, se = one(False)
    lg, gg, sg = one(True)
    lg2, gg2, _ = one(True)
    rel_g, rel_g2 = float((ge_ - gg).norm() / ge_.norm()), float((gg2 - gg).norm() / gg.norm())
    rel_s = float((se - sg).norm() / se.norm())
    print(f"[segments] loss eager {le:.7f} graph {lg:.7f}; gradient rel eager/graph {rel_g:.2e}, graph/graph {rel_g2:.2e}; forward state rel {rel_s:.2e}")
    assert abs(le - lg) <= 1e-6 * abs(le) and lg2 == lg          # forward: reproducible
    assert rel_s <= 1e-6                                          # observers and running statistics after the step
    assert rel_g <= 2e-2 and rel_g2 <= 2e-2                       # measured 1e-7 ... 3e-3 (one flipped rounding, amplified by the S1/S2 cancellation)
    # and the replayed gradients drive the optimizer
    p0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    plan = opt.prepare_step()
    seg.replay()
    opt.launch(plan)
    torch.cuda.synchronize()
    p1 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert float((p1 - p0).norm()) > 0 and bool(torch.isfinite(p1).all())


def test_bench_gpus2_spawns_two_ranks():
    """The launcher contract: `python bench.py --gpus 2` (no torchrun around it) must run TWO ranks and report n_gpus = 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--batch", "32", "--steps", "3", "--warmup", "2",
           "--check-allreduce", "--no-roofline", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 64 and rec["config"]["hip_graph"] is True
    ga = rec["config"]["grad_allreduce"]
    assert "buckets overlapped" in ga["mode"] and ga["fallback"] is None and len(ga["buckets"]) >= 2 and rec["value"] > 0
    chk = ga["allreduce_check"]        # the live two-rank reduction leaves the mean of the ranks' own gradients (and one rank's gradient alone is far from it)
    assert chk["ok"] and chk["single_rank_vs_mean"] > 10 * chk["rel_err_vs_mean_of_rank_gradients"], chk


def test_bench_two_physical_gpus_rccl():
    """first contact with more than one physical GPU must be boring.  When >= 2 devices are visible, `python bench.py --gpus 2` (RCCL, one rank
    per GPU, no --share-gpu) must come up on distinct devices, capture the segmented step without falling back, and leave the MEAN of the two ranks'
    own shard gradients in the arena (checked on the live process group by --check-allreduce); the exposed communication per step is recorded."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (the round-end 8-GPU node); the one-GPU boxes run the same path as --share-gpu / --force-dp")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "5", "--warmup", "3", "--check-allreduce",
           "--no-roofline", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    ga = rec["config"]["grad_allreduce"]
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 128
    assert ga["fallback"] is None and "buckets overlapped" in ga["mode"], ga
    chk = ga["allreduce_check"]
    assert chk["ok"] and sorted(chk["devices"]) == [0, 1] and chk["single_rank_vs_mean"] > 10 * chk["rel_err_vs_mean_of_rank_gradients"], chk
    assert ga["exposed_comm_ms_per_step"] is not None and rec["value"] > 0
    print(f"[2 GPUs] {rec['value']:.0f} img/s, exposed communication {ga['exposed_comm_ms_per_step']} ms / step, all-reduce check {chk}")


def _bench_ranks(extra, timeout=1800, **more_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **more_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra + ["--no-roofline", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    if out.returncode == 3 and "process-group bring-up failed" in out.stderr:
        # bench.py's own exit code for a rendezvous that did not come up (eight fresh interpreters importing torch beside a loaded parent): infrastructure, not the
        # path under test -- one more attempt; anything else (a capture fallback, a failed reduction check, a kernel error) fails right away
        import time
        time.sleep(5)
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_eight_rank_rehearsal_classifier_on_one_gpu():
    """the 8-GPU launch rehearsed on ONE device -- eight processes (gloo), each with its own replica, shard and captured chain of
    hipGraph segments; ports, memory, the bucket plan and the live eight-rank reduction (mean of the ranks' own gradients) all have to work, and a
    capture fallback would end the run (no --allow-fallback)."""
    rec = _bench_ranks(["--gpus", "8", "--share-gpu", "--batch", "8", "--steps", "3", "--warmup", "2", "--check-allreduce"])
    ga = rec["config"]["grad_allreduce"]
    assert rec["n_gpus"] == 8 and rec["config"]["global_batch"] == 64 and rec["config"]["hip_graph"] is True
    assert ga["fallback"] is None and "buckets overlapped" in ga["mode"] and len(ga["buckets"]) >= 2
    chk = ga["allreduce_check"]
    assert chk["ok"] and chk["single_rank_vs_mean"] > 10 * chk["rel_err_vs_mean_of_rank_gradients"], chk


def test_eight_rank_rehearsal_detector_on_one_gpu():
    """BASELINE.json config c5 as data parallel (Object_Detection/qtrainval.py:123-127): `bench.py --workload detect --gpus 8`, eight ranks on one device.
    The detector's step body is maps -> MultiBoxLoss -> hand-written backward with the same bucketed exchange (a bucket may be several arena ranges:
    `loc` / `conf` are registered as two ModuleLists while the backward visits them interleaved; with four buckets they coalesce into one range each)."""
    rec = _bench_ranks(["--workload", "detect", "--gpus", "8", "--share-gpu", "--batch", "4", "--res", "256", "--steps", "3", "--warmup", "2", "--check-allreduce"])
    ga = rec["config"]["grad_allreduce"]
    assert rec["n_gpus"] == 8 and rec["config"]["global_batch"] == 32 and rec["config"]["hip_graph"] is True and "SSDLite" in rec["metric"]
    assert ga["fallback"] is None and "buckets overlapped" in ga["mode"] and len(ga["buckets"]) >= 2
    chk = ga["allreduce_check"]
    assert chk["ok"] and chk["single_rank_vs_mean"] > 10 * chk["rel_err_vs_mean_of_rank_gradients"], chk


def test_two_rank_data_parallel_in_the_fp32_gradient_mode():
    """The fp32-gradient mode (the reference's gradient precision, Classification/utils/helper_functions.py:139-143; csrc/frost_g32.hip) under the same data-parallel
    machinery: two ranks on one device, captured segments, bucketed exchange -- the arena (fp32 either way) must hold the mean of the ranks' own gradients."""
    rec = _bench_ranks(["--gpus", "2", "--share-gpu", "--batch", "8", "--steps", "2", "--warmup", "2", "--check-allreduce"], FROST_GRAD="fp32")
    ga = rec["config"]["grad_allreduce"]
    assert rec["n_gpus"] == 2 and rec["config"]["grad_dtype"] == "fp32" and rec["config"]["hip_graph"] is True
    assert ga["fallback"] is None and ga["allreduce_check"]["ok"] and ga["allreduce_check"]["single_rank_vs_mean"] > 0.1, ga


def test_detector_segmented_step_equals_module_surface():
    """SegmentedStep's detector body (maps -> loss -> Engine.backward with bucket boundaries) leaves the same gradients in the arena as the module
    surface (model(x) -> MultiBoxLoss -> loss.backward()), eager and as a replayed chain of segments; and the bucket plan tiles the arena."""
    from frostnet_amd import frostnet as F, ssdlite as S
    from frostnet_amd.parallel import SegmentedStep
    torch.manual_seed(0)
    model = S.SSDLiteFrostNet(num_classes=21, mode="small", cfg=S.ssd_cfg_for(256))
    F.qat_prepare(model, version=0)
    model.cuda().train()
    runner = model.hip_runner()
    x = torch.randn(4, 3, 256, 256, device="cuda")
    boxes = [torch.tensor([[0.2, 0.2, 0.6, 0.7, 3.0]], device="cuda"), torch.tensor([[0.1, 0.3, 0.5, 0.9, 7.0], [0.5, 0.5, 0.9, 0.8, 1.0]], device="cuda"),
             torch.tensor([[0.3, 0.1, 0.8, 0.6, 11.0]], device="cuda"), torch.tensor([[0.4, 0.4, 0.7, 0.7, 19.0]], device="cuda")]
    tgt = S.pad_targets(boxes, torch.device("cuda"))
    mbl = S.MultiBoxLoss(21)

    def maps_loss(maps, tt):
        ll, lc = mbl(model._assemble(maps), tt)
        return ll + lc
    seg = SegmentedStep(runner, nbuckets=4, maps_loss=maps_loss)
    ranges = sorted(r for rs in seg.bucket_ranges() for r in rs)
    assert ranges[0][0] == 0 and ranges[-1][1] == runner.grad_arena.numel() and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert len(seg.cuts) >= 3
    model(x)                                       # tables / observers exist
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def surface():
        model.load_state_dict(sd0)
        for p in model.parameters():
            p.grad = None
        ll, lc = mbl(model(x), tgt)
        (ll + lc).backward()
        torch.cuda.synchronize()
        return float(ll + lc), runner.grad_arena.clone()

    def segmented(graph):
        model.load_state_dict(sd0)
        loss = seg.replay() if graph else seg.run_eager(x, tgt)
        torch.cuda.synchronize()
        return float(loss), runner.grad_arena.clone()
    l0, g0 = surface()
    l1, g1 = segmented(False)
    seg.capture(x, tgt)
    assert len(seg.graphs) == len(seg.cuts)
    l2, g2 = segmented(True)
    r1, r2 = float((g1 - g0).norm() / g0.norm()), float((g2 - g0).norm() / g0.norm())
    print(f"[detector segments] loss {l0:.6f} / {l1:.6f} / {l2:.6f}; gradient rel surface vs eager segments {r1:.2e}, vs replayed {r2:.2e}")
    assert abs(l1 - l0) <= 1e-5 * abs(l0) and abs(l2 - l0) <= 1e-5 * abs(l0)
    assert r1 <= 2e-2 and r2 <= 2e-2              # two runs of the same backward differ by the order of their fp32 atomics (see the classifier test above)
