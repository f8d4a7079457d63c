"""CPU-only tests (run with -m "not gpu"): C-ABI exports, nn.Module / optimizer / harness surface parity with the
reference (golden vectors), data-parallel gradient sync over gloo.  No compute call touches the HIP library here."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from oracle import frost_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import frostnet_amd
    return frostnet_amd


def test_cabi_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "frost_hip.h")).read()
    declared = set(re.findall(r"\b(frost_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = ctypes.CDLL(built.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/frost_hip.h but not exported"
    assert set(built.SYMBOLS) == declared          # the ctypes binding covers exactly the header
    assert built.load_library().frost_abi_version() == 5 and built.load_library().frost_ticket_words() == 40


def test_no_cpu_fallback_on_device_path(built):
    from frostnet_amd import frostnet as F
    from frostnet_amd.optimizer import QSGD
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        QSGD([p], lr=0.1).step()                   # GradBoost kernels are HIP-only: CPU tensors fail loudly
    m = F.frostnet_small_1_0()
    assert not m._is_qat_prepared()


def test_factories_and_state_dict_layout(built, golden):
    from frostnet_amd import frostnet as F
    g = golden("g7_scalars")
    names = [f"frostnet_{q}{m}_{t}" for q in ("quant_", "") for m in ("large", "base", "small")
             for t in ("1_25", "1_0", "0_75", "0_5", "0_35")]
    assert sorted(names) == sorted(F.MODEL_REGISTRY) and len(names) == 30
    for mode in ("large", "base", "small"):
        for tag in ("1_0", "0_5", "1_25"):
            net = F.MODEL_REGISTRY[f"frostnet_quant_{mode}_{tag}"](drop_rate=0.0, num_classes=10, pretrained=False)
            assert list(net.state_dict().keys()) == [str(k) for k in g[f"{mode}_{tag}_float_keys"]]
            assert [n for n, _ in net.named_parameters()] == [str(k) for k in g[f"{mode}_{tag}_param_names"]]
            assert [p.numel() for p in net.parameters()] == g[f"{mode}_{tag}_param_numel"].tolist()
            if tag == "1_0":
                F.qat_prepare(net, version=0)
                assert list(net.state_dict().keys()) == [str(k) for k in g[f"{mode}_{tag}_qat_keys"]]
    net = F.create_model("frostnet_quant_large_1_0")
    from frostnet_amd.harness import make_param_groups
    groups = make_param_groups(net, 1.0)
    cnt = {0.0: 0, 1.0: 0, 0.01: 0}
    for gr in groups:
        cnt[gr["weight_decay"]] += gr["params"][0].numel()
    assert [cnt[0.0], cnt[1.0], cnt[0.01]] == g["large_group_counts"].tolist()
    assert len(groups) == 209 and sum(p.numel() for p in net.parameters()) == 5807056


def test_kaiming_init_parity_with_reference_seed(built, golden):
    import zlib
    from frostnet_amd import frostnet as F
    g = golden("g5_fp32_eval")
    torch.manual_seed(1882)
    net = F.frostnet_small_1_0(drop_rate=0.0)
    sd = net.state_dict()
    assert np.uint32(zlib.crc32(sd["conv1.conv.0.weight"].numpy().tobytes())) == g["init1882_small_conv1_crc"]
    assert np.uint32(zlib.crc32(sd["classifier.2.weight"].numpy().tobytes())) == g["init1882_small_cls_crc"]


def test_config_c1_small_fp32_forward_cpu(built, golden):
    """BASELINE.json configs[0]: FrostNet-Small fp32 forward-only, batch=1, 224x224 on CPU (plumbing)."""
    from frostnet_amd import frostnet as F
    g = golden("g5_fp32_eval")
    B, res, seed, wseed = [int(v) for v in g["fp32_eval_small_spec"]]
    net = F.frostnet_small_1_0(drop_rate=0.0)
    spec = O.float_state_spec(O.net_cfg("small", 1.0))
    net.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], wseed))
    net.eval()
    with torch.no_grad():
        y = net(T(O.synth((B, 3, res, res), seed)))
    np.testing.assert_allclose(y.numpy(), g["fp32_eval_small_logits"], rtol=1e-5, atol=1e-5)


def test_qat_cpu_module_path_matches_reference(built, golden):
    """The stock-module (CPU) QAT graph of frostnet_amd.FrostNet reproduces the reference's QAT step (golden G5)."""
    from frostnet_amd import frostnet as F
    g = golden("g5_qat_large")
    B, res, seed, wseed = [int(v) for v in g["spec"]]
    torch.set_num_threads(8)
    net = F.frostnet_quant_large_1_0(drop_rate=0.0)
    spec = O.float_state_spec(O.net_cfg("large", 1.0))
    net.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], wseed))
    F.qat_prepare(net, version=0)
    y = net(T(O.synth((B, 3, res, res), seed)))
    np.testing.assert_allclose(y.detach().numpy(), g["s0_logits"], rtol=1e-5, atol=1e-5)


def test_lr_schedule_and_get_optimizer(built, golden):
    from frostnet_amd import harness, optimizer
    g = golden("g7_scalars")

    class A:
        anneal, epochs, warmup_epochs, warmup_lr, lr, restart_epochs = False, 400, 5, 0.0, 5e-3, 100
        learning_rate, weight_decay, nesterov, clip_by, toss_coin, noise_decay, amsgrad = 5e-3, 1e-5, True, 1e-3, True, 1e-2, False

    class Opt:
        param_groups = [dict(lr=0.0)]
    for (ep, it), lr in zip(g["lr_points"], g["lr_values"]):
        assert harness.adjust_learning_rate_cosine(Opt(), int(ep), int(it), 10, A) == pytest.approx(float(lr), rel=1e-12, abs=1e-18)
    p = [torch.nn.Parameter(torch.zeros(3))]
    for name, cls in (("SGD", torch.optim.SGD), ("RMS", torch.optim.RMSprop), ("Adam", torch.optim.Adam), ("AdamW", torch.optim.AdamW),
                      ("QSGD", optimizer.QSGD), ("QRMS", optimizer.QRMSprop), ("QAdam", optimizer.QAdam), ("QAdamW", optimizer.QAdamW)):
        o = optimizer.get_optimizer(name, p, A)
        assert type(o) is cls
        if name.startswith("Q"):
            assert o.is_warmup is True and o.defaults["clip_by"] == 1e-3
    out = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1]])
    assert [float(a) for a in harness.accuracy(out, torch.tensor([1, 2]), topk=(1, 2))] == [50.0, 50.0]


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from frostnet_amd.parallel import GradSync, broadcast_model
    sizes = [7, 1000, 13, 2048, 64, 5]
    offs = np.cumsum([0] + sizes[:-1]).tolist()
    arena = torch.zeros(sum(sizes))
    sync = GradSync(arena, offs, nbuckets=3)
    torch.manual_seed(100 + rank)
    grads = [torch.randn(s) for s in sizes]
    for i in reversed(range(len(sizes))):           # backward order = reverse parameter order
        arena[offs[i]: offs[i] + sizes[i]] = grads[i]
        sync.ready(offs[i])
    sync.finish()
    lin = torch.nn.Linear(3, 2)
    # buffers of every dtype a QAT model carries (int64 num_batches_tracked, uint8 observer switches, a bool, a non-contiguous float view, a 0-d tensor): the
    # broadcast is coalesced per dtype and must put each value back where it belongs
    lin.register_buffer("nbt", torch.tensor(3 + rank, dtype=torch.int64))
    lin.register_buffer("flags", torch.tensor([rank, 1 - rank, 1], dtype=torch.uint8))
    lin.register_buffer("on", torch.tensor([bool(rank), True]))
    lin.register_buffer("view", torch.arange(12, dtype=torch.float32).reshape(3, 4).t()[:, 1:] * (1 + rank))
    torch.manual_seed(rank)
    with torch.no_grad():
        lin.weight.normal_()
    broadcast_model(lin)
    extra = [lin.nbt.item(), lin.flags.tolist(), lin.on.tolist(), lin.view.tolist(), lin.bias.detach().tolist()]
    q.put((rank, arena.numpy().copy(), lin.weight.detach().numpy().copy(), len(sync.buckets), extra))
    dist.destroy_process_group()


def test_data_parallel_grad_sync_gloo_world2():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    sizes = [7, 1000, 13, 2048, 64, 5]
    expect = []
    for s_i, s in enumerate(sizes):
        gs = []
        for r in range(2):
            torch.manual_seed(100 + r)
            gs.append([torch.randn(x) for x in sizes][s_i])
        expect.append((gs[0] + gs[1]) / 2)
    expect = torch.cat(expect)
    assert res[0][3] >= 2
    for r in range(2):
        torch.testing.assert_close(T(res[r][1]), expect)          # every rank holds the mean of the per-shard gradients
    torch.testing.assert_close(T(res[0][2]), T(res[1][2]))           # broadcast made the replicas identical
    assert res[0][4] == res[1][4] and res[1][4][0] == 3 and res[1][4][1] == [0, 1, 1] and res[1][4][2] == [False, True]          # ... rank 0's values, every dtype
    assert res[1][4][3] == (torch.arange(12, dtype=torch.float32).reshape(3, 4).t()[:, 1:]).tolist()


def test_features_backbone_cpu_matches_reference(built, golden):
    """frostnet_features surface: keys, four maps [x1,x2,x3,x5], values vs the reference golden (G8)."""
    from frostnet_amd import frostnet_features as FF
    g = golden("g8_features")
    B, res, seed, wseed = [int(v) for v in g["spec"]]
    net = FF.FrostNet(mode="large", width_mult=1.0)
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    spec = O.float_state_spec(O.net_cfg("large", 1.0), features=True)
    net.load_state_dict(O.synth_state([k for k, _ in spec], [s for _, s in spec], wseed))
    net.eval()
    with torch.no_grad():
        fs = net(T(O.synth((B, 3, res, res), seed)))
    assert len(fs) == 4
    for i, f in enumerate(fs):
        assert list(f.shape) == g[f"f{i}_shape"].tolist()
        np.testing.assert_allclose(f[0, :8, :4, :4].numpy(), g[f"f{i}_crop"], rtol=1e-4, atol=1e-5)
    assert [f.shape[1] for f in fs] == [24, 40, 96, 320]


def test_bench_launcher_dry_run_spawns_n_ranks():
    """`python bench.py --gpus N` must start N ranks itself (VERDICT r1 missing #1).  CPU dry run: N gloo ranks, one all-reduce."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run-launcher"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 3 and rec["allreduce_sum"] == rec["expected"] == 6.0
    # launched BY a launcher (RANK / WORLD_SIZE already set): no second spawn, world size taken from the environment
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29777")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run-launcher"], capture_output=True,
                         text=True, timeout=300, env=env)
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1


def test_bench_launcher_dry_run_eight_ranks_and_failure_modes():
    """the node-sized launch (8 ranks: spawn, rendezvous on 127.0.0.1, one all-reduce, ONE line from rank 0), a world size that
    contradicts --gpus is refused with a message instead of silently re-labelled, and a rendezvous that cannot complete ends with exit code 3
    after the hard timeout instead of hanging."""
    import json
    import subprocess
    import time
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-launcher"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["allreduce_sum"] == rec["expected"] == 36.0
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29779")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run-launcher"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 2 and "WORLD_SIZE=2 but --gpus 4" in out.stderr
    # rank 0 of a 2-rank world whose peer never shows up: the bring-up times out (FROST_RDZV_TIMEOUT) and the rank exits with code 3
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29781", FROST_RDZV_TIMEOUT="5")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-launcher"], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 3 and "process-group bring-up failed" in out.stderr, (out.returncode, out.stderr[-500:])
    assert time.time() - t0 < 200


def _shard_oracle_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from frostnet_amd.parallel import GradSync
    torch.set_num_threads(4)
    cfg = O.net_cfg("small", 1.0)
    P, Bf = O.make_state(O.float_state_spec(cfg), 5000, True)          # identical weights on every rank
    qs = O.QState(Bf)
    x = T(O.synth((2, 3, 32, 32), 60 + rank))                           # this rank's shard
    tgt = torch.tensor([1 + rank, 7 + rank])
    torch.nn.functional.cross_entropy(O.frostnet_forward(P, qs, cfg, x, True, True), tgt).backward()
    sizes = [p.numel() for p in P.values()]
    offs = np.cumsum([0] + sizes[:-1]).tolist()
    arena = torch.zeros(sum(sizes))
    sync = GradSync(arena, offs, nbuckets=4)
    for i, p in reversed(list(enumerate(P.values()))):                  # gradients become final in reverse parameter order
        arena[offs[i]: offs[i] + sizes[i]] = p.grad.reshape(-1)
        sync.ready(offs[i])
    sync.finish()
    q.put((rank, arena.numpy().copy()))
    dist.destroy_process_group()


def test_data_parallel_shard_oracle_mean_gradient_gloo_world2():
    """SURVEY 8(e) parity check without a cluster: W=2 shards through the oracle from identical weights -> the mean gradient must be what
    every rank holds after the bucketed all-reduce (real FrostNet-Small gradients in real parameter order, not random vectors)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 300)
    procs = [ctx.Process(target=_shard_oracle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    cfg = O.net_cfg("small", 1.0)
    grads = []
    torch.set_num_threads(4)
    for r in range(2):
        P, Bf = O.make_state(O.float_state_spec(cfg), 5000, True)
        qs = O.QState(Bf)
        torch.nn.functional.cross_entropy(O.frostnet_forward(P, qs, cfg, T(O.synth((2, 3, 32, 32), 60 + r)), True, True),
                                          torch.tensor([1 + r, 7 + r])).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in P.values()]))
    expect = ((grads[0] + grads[1]) / 2).numpy()
    for r in range(2):
        np.testing.assert_allclose(res[r][1], expect, rtol=1e-5, atol=1e-7)


def test_ssd_priorbox_and_multibox_loss_match_reference(built, golden):
    """PriorBox and MultiBoxLoss of frostnet_amd.ssdlite (vectorised restatements) against the reference's own classes
    (Object_Detection/layers/functions/prior_box.py, layers/modules/multibox_loss.py; fixture tools/gen_golden.py g11)."""
    import zlib
    from frostnet_amd import ssdlite as S
    g = golden("g11_detection")
    pri = S.prior_boxes(S.SSD512_VOC)
    assert list(pri.shape) == g["priors_shape"].tolist()
    assert np.uint32(zlib.crc32(pri.numpy().tobytes())) == g["priors_crc"]            # bit-exact
    crit = S.MultiBoxLoss(21, 0.5, 3, (0.1, 0.2))
    P = pri.shape[0]
    for case in range(2):
        loc = (T(O.synth((3, P, 4), 1100 + case)) * 0.5).requires_grad_(True)
        conf = (T(O.synth((3, P, 21), 1110 + case)) * (1.0 + case)).requires_grad_(True)
        tg = [T(g[f"c{case}_t{n}"]) for n in range(3)]
        ll, lc = crit((loc, conf, pri), tg)
        (ll + lc).backward()
        np.testing.assert_allclose([float(ll), float(lc)], g[f"c{case}_losses"], rtol=2e-6)
        for name, t in (("dloc", loc.grad), ("dconf", conf.grad)):
            pack = g[f"c{case}_{name}"]
            mine = O.sample_big(t.double().numpy())
            np.testing.assert_allclose(mine, pack[3:], rtol=1e-5, atol=1e-9)
            np.testing.assert_allclose(np.abs(t.double().numpy()).sum(), pack[1], rtol=1e-6)


def test_ssdlite_module_surface(built):
    from frostnet_amd import ssdlite as S, frostnet as F
    m = S.SSDLiteFrostNet(num_classes=21, mode="small")
    assert m.source_channels == [40, 96, 320, 512, 256, 256] and m.priors.shape == (24528, 4)
    F.qat_prepare(m, version=0)
    loc, conf, pri = m(torch.randn(2, 3, 128, 128))
    assert loc.shape == (2, 1536, 4) and conf.shape == (2, 1536, 21)
    keys = list(m.state_dict().keys())
    assert "extras.0.dw.conv.0.weight_fake_quant.scale" in keys and "conf.5.pw.conv.0.bn.running_var" in keys
    assert "priors" not in keys
