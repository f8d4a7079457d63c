"""Headline benchmark: images/sec, FrostNet-Large 224x224 QAT forward+backward (+GradBoost step) on N MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: spawns N ranks itself via torch.distributed.run, or runs as a rank when
                                                          the launcher's RANK / WORLD_SIZE are already in the environment)

One "step" = one pass of the hot path over one synthetic batch: fake-quantised forward (int8 MFMA / LDS depthwise),
hand-written backward, GradBoost-SGD multi-tensor update; for N>1 plus the bucketed RCCL gradient all-reduce
overlapped with backward.  Per-GPU batch is fixed (weak scaling).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per image, FrostNet-Large w=1.0 @224, int8 activations + bf16 gradients
ALGO_BYTES_PER_IMG = 45_593_016
FLOPS_I8_PER_IMG, FLOPS_BF16_PER_IMG = 861626432, 1701576832          # forward convs; dgrad + wgrad
FLOPS_PER_IMG = FLOPS_I8_PER_IMG + FLOPS_BF16_PER_IMG                        # 2 563 203 264
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable


def cpu_baseline(res=224, batch=64, timed=3):
    """Reference-path stand-in timed on the host cores: the CPU oracle (restated torch eager QAT graph, kind 'port').
    Bounded sample per SURVEY 8(d): batch 64, ONE warm-up step + THREE timed steps of fwd + bwd + GradBoost-SGD on the same state, median reported
    (~25 s of CPU work on the GPU box's 32 usable threads; FROST_CPU_BASELINE_BATCH overrides the batch on small hosts)."""
    from oracle import frost_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))          # more intra-op threads than usable cores makes the torch CPU kernels collapse
    torch.set_num_threads(cores)
    batch = int(os.environ.get("FROST_CPU_BASELINE_BATCH", batch))
    cfg = O.net_cfg("large", 1.0)
    hp = dict(lr=5e-3, momentum=0.9, weight_decay=1e-5, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2)
    P, B = O.make_state(O.float_state_spec(cfg), 5000, True)
    qs = O.QState(B)
    x = torch.from_numpy(O.synth((batch, 3, res, res), 77))
    tgt = torch.randint(0, 1000, (batch,))
    states = {k: {} for k in P}

    def one_step():
        t0 = time.time()
        for p in P.values():
            p.grad = None
        y = O.frostnet_forward(P, qs, cfg, x, True, True)
        torch.nn.functional.cross_entropy(y, tgt).backward()
        with torch.no_grad():
            for k, p in P.items():
                O.gradboost_step("QSGD", p, p.grad, states[k], dict(hp, weight_decay=O.param_group_rule(tuple(p.shape), 1e-5)),
                                 boost=True, noise=torch.empty_like(p).exponential_(), coin=torch.randint(0, 2, p.shape).float())
        return time.time() - t0

    warm = one_step()
    ts = sorted(one_step() for _ in range(timed))
    dt = ts[len(ts) // 2]
    return dict(value=batch / dt, unit="images/sec", cores=cores, kind="port",
                sample=f"batch {batch} @ {res}x{res}, 1 warm-up + {timed} timed steps (median {dt:.2f} s; all {[round(t, 2) for t in ts]}, warm-up {warm:.2f} s), "
                       f"FrostNet-Large QAT fwd+bwd+GradBoost-SGD, torch {torch.__version__} CPU kernels via oracle/frost_oracle.py")


PMC_TAGS = ("r06", "r05", "r04", "r03", "r02")


def extra_leg(extra_args, extra_env=None, timeout=200):
    """One more single-GPU measurement of THIS script in a child process (its own device context: a failure or a stalled RCCL bring-up there costs this
    run a field, not its headline).  Returns the child's JSON line as a dict, or {"error": ...}."""
    import subprocess
    env = dict(os.environ)
    env.update(extra_env or {})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-roofline", "--no-extras"] + list(extra_args)
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, stdin=subprocess.DEVNULL)
        lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return dict(error=f"exit code {r.returncode}: {r.stderr.decode(errors='replace')[-300:]}")
        return json.loads(lines[-1])
    except Exception as e:  # pragma: no cover
        return dict(error=f"{type(e).__name__}: {e}")


def tape_algorithmic_bytes(tape, image_elems):
    """Algorithmic HBM bytes of one fwd + bwd pass from the engine's tape of a training forward, by SURVEY 8(d)'s layer-granular rule: every conv reads its input
    and writes its output once at 1 B / element (forward), its backward reads dY (2 B, bf16) and the saved input (1 B) and writes dX (2 B) unless the input
    is the image.  `image_elems`: elements of the network input (the stem's input in the tape is its im2col view).  FrostNet-Large @224: 45 593 016 B / image."""
    total = 0
    for e in tape:
        if e[0] == "conv":
            _, l, x, y = e
            xin = image_elems if l.kind == "stem" else x.numel
            total += xin + y.numel + 2 * y.numel + xin + (0 if l.kind == "stem" else 2 * xin)
        elif e[0] == "head":
            _, l, x = e[:3]
            xin, yout = x.n * l.cin_g, x.n * l.cout          # the classifier conv on the pooled vector
            total += xin + yout + 2 * yout + xin + 2 * xin
    return total


def pmc_table(batch):
    """The committed PMC family table of this same step (profiles/r0N_kernels_b<batch>.json, made by tools/collect_profiles.sh: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs, gfx950 correction applied).  Counters cannot be collected from inside this process, so traffic numbers are the
    committed ones or null.  Returns (document, relative path) or (None, None)."""
    here = os.path.dirname(os.path.abspath(__file__))
    for tag in PMC_TAGS:
        path = os.path.join(here, "profiles", f"{tag}_kernels_b{batch}.json")
        try:
            return json.load(open(path)), os.path.relpath(path, here)
        except (OSError, ValueError):
            continue
    return None, None


def pmc_traffic(label, batch):
    """HBM bytes per launch of one kernel family from the committed PMC passes."""
    doc, path = pmc_table(batch)
    try:
        return int(doc["families"][label]["hbm_bytes_per_launch"]), path
    except (TypeError, KeyError, ValueError):
        return None, None


def pmc_step_traffic(batch):
    """Whole-step HBM bytes (sum over every kernel family of the committed table) and the launches per step of that pass."""
    doc, path = pmc_table(batch)
    if doc is None:
        return None, None, None
    if "hbm_bytes_per_step" in doc:
        return int(doc["hbm_bytes_per_step"]), doc.get("kernel_launches_per_step"), path
    fams = doc.get("families", {})
    steps = None
    if "gradboost" in fams and fams["gradboost"].get("FETCH_SIZE_launches"):
        steps = fams["gradboost"]["FETCH_SIZE_launches"]            # one optimizer launch per step of the counter pass
    if not steps:
        return None, None, path
    tot = sum(v["hbm_bytes_per_launch"] * v.get("FETCH_SIZE_launches", 0) / steps for v in fams.values() if "hbm_bytes_per_launch" in v)
    n = sum(v.get("FETCH_SIZE_launches", 0) / steps for k, v in fams.items() if k != "torch_elementwise")
    return int(tot), round(n, 1), path


def graph_census(graphs):
    """Node count of the captured step, read from the hipGraph objects themselves (hipGraphGetNodes / hipGraphNodeGetType / hipGraphKernelNodeGetParams +
    hipKernelNameRefByPtr): total nodes, kernel nodes split into this library's (k_* / frost_*) and everything else (aten element-wise kernels), memset and
    memcpy nodes.  `graphs`: torch.cuda.CUDAGraph objects captured with keep_graph=True.  None if the runtime does not expose the graph."""
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")

        class Dim3(C.Structure):
            _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]

        class KParams(C.Structure):
            _fields_ = [("blockDim", Dim3), ("extra", C.c_void_p), ("func", C.c_void_p), ("gridDim", Dim3), ("kernelParams", C.c_void_p), ("sharedMemBytes", C.c_uint)]
        hip.hipKernelNameRefByPtr.restype = C.c_char_p
        hip.hipKernelNameRefByPtr.argtypes = [C.c_void_p, C.c_void_p]
        out = dict(nodes_total=0, kernels_own=0, kernels_other=0, memset=0, memcpy=0, other_nodes=0, other_kernel_names={})
        for g in graphs:
            raw = C.c_void_p(g.raw_cuda_graph())
            n = C.c_size_t(0)
            if hip.hipGraphGetNodes(raw, None, C.byref(n)) != 0:
                return None
            nodes = (C.c_void_p * n.value)()
            if hip.hipGraphGetNodes(raw, nodes, C.byref(n)) != 0:
                return None
            for nd in nodes:
                t = C.c_int(-1)
                hip.hipGraphNodeGetType(C.c_void_p(nd), C.byref(t))
                out["nodes_total"] += 1
                if t.value == 0:
                    kp = KParams()
                    name = b""
                    if hip.hipGraphKernelNodeGetParams(C.c_void_p(nd), C.byref(kp)) == 0 and kp.func:
                        name = hip.hipKernelNameRefByPtr(kp.func, None) or b""
                    nm = name.decode(errors="replace")
                    import re
                    if re.match(r"^(_Z\d+)?(k_|frost_)", nm):      # this library's kernels are global-namespace k_* (mangled: _Z<len>k_...)
                        out["kernels_own"] += 1
                    else:
                        out["kernels_other"] += 1
                        key = nm[:60] or "?"
                        out["other_kernel_names"][key] = out["other_kernel_names"].get(key, 0) + 1
                elif t.value == 1:
                    out["memcpy"] += 1
                elif t.value == 2:
                    out["memset"] += 1
                else:
                    out["other_nodes"] += 1
        return out
    except Exception as e:  # pragma: no cover
        return dict(error=f"{type(e).__name__}: {e}")


def side_workload(args, dev):
    """The two other device workloads, same timing protocol, single GPU: `infer` = BASELINE.json config c2 (float model, bf16 inference,
    B = 256 by default), `float` = the StatAssist warm-up training step (float model, forward + backward + QSGD step, is_warmup)."""
    import __graft_entry__ as ge
    ge.build()
    from frostnet_amd import frostnet as F, harness as H
    from frostnet_amd.optimizer import QSGD
    torch.manual_seed(1882)
    batch = args.batch if args.batch != 512 else 256
    g = torch.Generator(device=dev).manual_seed(1882)
    if args.workload in ("infer", "float"):
        model = F.MODEL_REGISTRY[f"frostnet_{args.mode}_1_0"]().to(dev)
        x = torch.randn(batch, 3, args.res, args.res, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        tgt = torch.randint(0, 1000, (batch,), device=dev, generator=g)
    if args.workload == "detect":
        from frostnet_amd import ssdlite as S
        batch = args.batch if args.batch != 512 else 32
        res = args.res if args.res != 224 else 512
        model = S.SSDLiteFrostNet(num_classes=21, mode=args.mode)
        F.qat_prepare(model, version=0)
        model.to(dev).train()
        x = torch.randn(batch, 3, res, res, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        rng = torch.Generator().manual_seed(1882)
        tgts = []
        for i in range(batch):
            k = 1 + i % 4
            c, wh = torch.rand(k, 2, generator=rng) * 0.5 + 0.25, torch.rand(k, 2, generator=rng) * 0.3 + 0.1
            tgts.append(torch.cat([c - wh / 2, c + wh / 2, torch.randint(0, 20, (k, 1), generator=rng).float()], 1).to(dev))
        opt = QSGD(H.make_param_groups(model, 1e-5), lr=5e-3, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2)
        opt.is_warmup = False
        crit = S.MultiBoxLoss(21)
        tgts = S.pad_targets(tgts, dev)          # the collate step of a detection data loader: ragged boxes -> [N, K, 5] + validity mask

        def body():
            opt.zero_grad(set_to_none=True)
            loc, conf, pri = model(x)
            ll, lc = crit((loc, conf, pri), tgts)
            (ll + lc).backward()
        step = None
        # algorithmic bytes of the detector step by the headline's rule, counted from the engine's tape of one training forward: backbone
        # (Large @512: 2 139 234 304 MAC) + extras + the twelve prediction heads, every conv 1 B / element forward, bf16 gradients backward
        opt.zero_grad(set_to_none=True)
        loc0, conf0, pri0 = model(x)
        bytes_per_img = tape_algorithmic_bytes(model.hip_runner().E.tape, x.numel()) // batch
        ll0, lc0 = crit((loc0, conf0, pri0), tgts)
        (ll0 + lc0).backward()
        opt.step()
        del loc0, conf0, pri0, ll0, lc0
        metric, dtype = f"images/sec SSDLite-FrostNet-{args.mode.capitalize()} {res}x{res} QAT fwd+bwd", "int8"
        what = (f"SSDLite on the FrostNet-{args.mode.capitalize()} backbone (frostnet_amd.ssdlite), int8 fake-quant QAT fwd+bwd + MultiBoxLoss + GradBoost-SGD step, "
                f"batch={batch}, {res}x{res} (BASELINE.json config c5, per GPU)")
        args.res = res
    elif args.workload == "int8":
        batch = args.batch if args.batch != 512 else 256
        model = F.MODEL_REGISTRY[f"frostnet_quant_{args.mode}_1_0"]()
        F.qat_prepare(model, version=0)
        model.to(dev).train()
        x = torch.randn(batch, 3, args.res, args.res, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            model(x)                         # one calibration forward (observers, BN statistics), as Classification/evaluate.py:104-112
        model.hip_convert()
        step = lambda: model(x)
        bytes_per_img, metric, dtype = 13_066_856, "images/sec FrostNet-Large 224x224 converted int8 inference", "int8"
        what = (f"FrostNet-{args.mode.capitalize()} converted int8 inference (torch.quantization.convert + QNNPACK semantics, bit-exact vs the reference), "
                f"batch={batch}, {args.res}x{args.res}")
    elif args.workload == "infer":
        model.eval()
        step = lambda: model.hip_infer_bf16(x)
        bytes_per_img, metric, dtype = 26_130_000, "images/sec FrostNet-Large 224x224 bf16 inference", "bf16"
        what = f"FrostNet-{args.mode.capitalize()} float model, bf16 inference (BatchNorm folded), batch={batch}, {args.res}x{args.res} NHWC (BASELINE.json config c2)"
    else:
        model.train()
        opt = QSGD(H.make_param_groups(model, 1e-5), lr=5e-3, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2)
        crit = torch.nn.CrossEntropyLoss()

        def body():                              # harness.train_one_iter without its optimizer step (helper_functions.py:139-142)
            opt.zero_grad(set_to_none=True)
            crit(model(x), tgt).backward()
        step = None
        prec = model.hip_runner().precision          # activation storage: bf16 (default) or fp32 (FROST_FLOAT_PRECISION=fp32, the reference's precision)
        bytes_per_img, metric, dtype = None, "images/sec FrostNet-Large 224x224 float warm-up fwd+bwd", prec
        what = (f"FrostNet-{args.mode.capitalize()} float model (StatAssist warm-up), {prec} activations, fwd+bwd + GradBoost-SGD step (is_warmup), "
                f"batch={batch}, {args.res}x{args.res}")
    graphed = False
    if step is None:                             # a training workload: zero_grad -> forward -> loss -> backward -> optimizer step, as ONE hipGraph like the headline
        def step():
            body()
            opt.step()
        for _ in range(3):                       # tables / optimizer state exist before the capture
            step()
        torch.cuda.synchronize()
        # (the float warm-up step is GPU-bound with its weight gradients on a second stream: 17.0 ms eager, 17.6 ms replayed (FROST_FLOAT_GRAPH=1) -- left eager)
        if not args.no_graph and (args.workload == "detect" or (args.workload == "float" and os.environ.get("FROST_FLOAT_GRAPH", "0") == "1")):
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    plan = opt.prepare_step()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side):
                        body()
                        opt.launch(plan)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()

                def step():
                    opt.prepare_step()           # host side of the optimizer step: counters and the device-resident scalars the captured launch reads
                    graph.replay()
                graphed = True
            except Exception as e:  # pragma: no cover
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                torch.cuda.synchronize()
    elif not args.no_graph and args.workload in ("infer", "int8"):
        # inference: the whole forward (static input buffer) replayed as ONE hipGraph -- the ~100 launches of a forward are 10-40 us each
        eager = step
        for _ in range(3):
            eager()
        torch.cuda.synchronize()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    static_out = eager()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            step = graph.replay
            graphed = True
        except Exception as e:  # pragma: no cover
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            step = eager
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = batch * args.steps / dt
    out = dict(metric=metric, value=round(value, 2), unit="images/sec", n_gpus=1, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling="weak", vs_baseline=None, dtype=dtype,
               data="synthetic", config=dict(workload=what, per_gpu_batch=batch, resolution=args.res, hip_graph=graphed))
    if bytes_per_img:
        gbs = value * bytes_per_img / 1e9
        out["roofline"] = dict(bound="hbm", kernel="whole step", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                               traffic=None, algorithmic_bytes_per_image=bytes_per_img, algorithmic_bytes_per_step=bytes_per_img * batch)
    print(json.dumps(out), flush=True)


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (the driver may also
    launch us that way itself; then RANK / WORLD_SIZE are already set and this is skipped)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def _init_group(backend, dev, rank, world, args):
    """Rendezvous + communicator bring-up with a HARD timeout and a readable failure: a stalled store or an RCCL init that never returns ends the
    run with one message per rank instead of hanging the node (FROST_RDZV_TIMEOUT seconds, default 300)."""
    import datetime
    import torch.distributed as dist
    tmo = int(os.environ.get("FROST_RDZV_TIMEOUT", "300"))
    kw = dict(timeout=datetime.timedelta(seconds=tmo))
    if dev is not None:
        kw["device_id"] = dev
    t0 = time.time()
    try:
        dist.init_process_group(backend, **kw)
        if backend == "nccl":                 # communicators are created lazily: force it now, inside the timeout, with a one-element collective
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != dist.get_world_size():
                raise RuntimeError(f"probe all-reduce returned {probe.item()} with world size {dist.get_world_size()}")
    except Exception as e:
        print(f"[bench] rank {rank}/{world}: process-group bring-up failed after {time.time() - t0:.0f} s (backend {backend}, MASTER_ADDR="
              f"{os.environ.get('MASTER_ADDR')}, MASTER_PORT={os.environ.get('MASTER_PORT')}, timeout {tmo} s): {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        sys.exit(3)
    if dist.get_world_size() != world:
        print(f"[bench] rank {rank}: process group has {dist.get_world_size()} ranks, WORLD_SIZE says {world}", file=sys.stderr, flush=True)
        sys.exit(3)


def _check_distinct_devices(dev, rank, world):
    """Every rank must drive its own GPU: gather (host, device index, PCI bus id / uuid) and refuse duplicates."""
    import socket
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(dev)
    ident = (socket.gethostname(), int(dev.index), str(getattr(props, "uuid", "")) or str(getattr(props, "pci_bus_id", "")))
    got = [None] * world
    dist.all_gather_object(got, ident)
    if len(set(got)) != world:
        if rank == 0:
            print(f"[bench] ranks do not see distinct devices: {got}", file=sys.stderr, flush=True)
        sys.exit(3)
    return got


def _rank_devices(dev, world):
    """[device index per rank] (all ranks call this)."""
    import torch.distributed as dist
    got = [None] * world
    dist.all_gather_object(got, int(dev.index))
    return got


def dry_run(args):
    """Launcher check without GPUs (CPU test / any box): N gloo ranks rendezvous, all-reduce a rank-dependent vector, rank 0 prints
    one line.  Exercises exactly the spawn + env + rendezvous + single-line-output plumbing of the N>1 path."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        _init_group("gloo", None, rank, world, args)
    t = torch.full((4,), float(rank + 1))
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(dict(metric="launcher dry run", n_gpus=world, requested=args.gpus, allreduce_sum=float(t[0]),
                              expected=world * (world + 1) / 2, dry_run=True)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="qat", choices=["qat", "infer", "float", "detect", "int8"],
                    help="qat (default): the headline metric, BASELINE.json config c3/c4; infer: config c2; float: the StatAssist warm-up step; "
                         "detect: config c5 (SSDLite-FrostNet 512x512 QAT fwd+bwd+step, per GPU); int8: converted int8 inference (SURVEY N2)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch (BASELINE.json config 3: 512)")
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--mode", default="large")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="N=1 headline run: skip the two extra legs (the fp32-gradient mode's step time; the data-parallel code path on a 1-rank RCCL group)")
    ap.add_argument("--buckets", type=int, default=4, help="N>1: gradient buckets = hipGraph segments of the backward pass")
    ap.add_argument("--single-allreduce", action="store_true",
                    help="N>1: one graph + ONE all-reduce after the backward instead of the bucketed, backward-overlapped default (A/B)")
    ap.add_argument("--force-dp", action="store_true", help="run the N>1 code path on a 1-rank process group (single-GPU check of that path)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="dev/test: all ranks on cuda:0 with the gloo backend (checks launcher + segmented step + a real 2-rank all-reduce on a 1-GPU box)")
    ap.add_argument("--dry-run-launcher", action="store_true", help="no GPU: N gloo ranks, one all-reduce, one JSON line")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N>1: if the segmented hipGraph capture fails, degrade to one graph + a single post-backward all-reduce (or eager launches) and still print a "
                         "line; WITHOUT this flag a failed capture ends the run with a non-zero exit code, so a degraded N-GPU number can never be the headline")
    ap.add_argument("--check-allreduce", action="store_true",
                    help="N>1: before timing, verify on the live process group that the bucketed all-reduce leaves the MEAN of the ranks' own shard gradients in every rank's arena")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(_self_launch(args))
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per requested GPU (torch.distributed.run --nproc-per-node {args.gpus})", file=sys.stderr, flush=True)
        sys.exit(2)
    if args.dry_run_launcher:
        return dry_run(args)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback on the product path)"
    if args.share_gpu:
        local_rank = 0
    assert local_rank < torch.cuda.device_count(), f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPUs visible"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dp = world > 1 or args.force_dp
    if args.workload != "qat" and not (args.workload == "detect" and dp):
        assert world == 1, "the side workloads infer / float / int8 are single-GPU measurements (detect runs data parallel: BASELINE.json config c5)"
        return side_workload(args, dev)
    import torch.distributed as dist
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        _init_group("gloo" if args.share_gpu else "nccl", None if args.share_gpu else dev, rank, world, args)
        if not args.share_gpu:
            _check_distinct_devices(dev, rank, world)
        if rank == 0:
            print(f"[bench] process group up: backend={dist.get_backend()} world={dist.get_world_size()} (one rank per GPU)", file=sys.stderr, flush=True)

    import __graft_entry__ as ge
    if local_rank == 0 and (rank == 0 or not args.share_gpu):
        ge.build()
    if world > 1:
        dist.barrier()
    import frostnet_amd
    from frostnet_amd import _lib as L
    from frostnet_amd import frostnet as F
    from frostnet_amd.optimizer import QSGD
    from frostnet_amd.parallel import SegmentedStep, broadcast_model

    torch.manual_seed(1882)                       # Classification/train.py:38-39
    detect = args.workload == "detect"
    g = torch.Generator(device=dev).manual_seed(1882 + rank)
    if detect:
        # BASELINE.json config c5: SSDLite on the FrostNet backbone, 512 x 512 QAT, data parallel (Object_Detection/qtrainval.py:123-127 wraps the net in
        # nn.DataParallel; here one process per GPU, the same bucketed gradient exchange as the classifier)
        from frostnet_amd import ssdlite as S, harness as H
        if args.batch == 512:
            args.batch = 32
        if args.res == 224:
            args.res = 512
        model = S.SSDLiteFrostNet(num_classes=21, mode=args.mode, cfg=S.ssd_cfg_for(args.res))
        F.qat_prepare(model, version=0)
        model.to(dev).train()
        broadcast_model(model)
        opt = QSGD(H.make_param_groups(model, 1e-5), lr=5e-3, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2)
        opt.is_warmup = False
        runner = model.hip_runner()
        x = torch.randn(args.batch, 3, args.res, args.res, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        rng = torch.Generator().manual_seed(1882 + rank)
        boxes = []
        for i in range(args.batch):
            k = 1 + i % 4
            c, wh = torch.rand(k, 2, generator=rng) * 0.5 + 0.25, torch.rand(k, 2, generator=rng) * 0.3 + 0.1
            boxes.append(torch.cat([c - wh / 2, c + wh / 2, torch.randint(0, 20, (k, 1), generator=rng).float()], 1).to(dev))
        tgt = S.pad_targets(boxes, dev)           # the collate step of a detection loader: ragged boxes -> [N, K, 5] + validity mask
        mbl = S.MultiBoxLoss(21)

        def loss_of(xx, tt):                      # module surface: model(x) -> (loc, conf, priors) -> MultiBoxLoss (qtrainval.py:186-190)
            ll, lc = mbl(model(xx), tt)
            return ll + lc

        def maps_loss(maps, tt):                  # the same loss from the twelve dequantised maps (SegmentedStep's body)
            ll, lc = mbl(model._assemble(maps), tt)
            return ll + lc
        seg_kw = dict(maps_loss=maps_loss)
    else:
        model = F.MODEL_REGISTRY[f"frostnet_quant_{args.mode}_1_0"]()     # drop_rate 0.2 active, as in training
        F.qat_prepare(model, version=0)
        model.to(dev).train()
        broadcast_model(model)
        wd = 1e-5                                     # Classification/setting/train.json:5-21
        groups = [{"params": [p], "weight_decay": (0.0 if p.shape[1] == 1 else wd) if p.dim() == 4 else wd * 0.01}
                  for p in model.parameters()]
        opt = QSGD(groups, lr=5e-3, momentum=0.9, nesterov=True, clip_by=1e-3, toss_coin=True, noise_decay=1e-2, weight_decay=wd)
        opt.is_warmup = False                         # StatAssist epoch done -> GradBoost noise on (train.py:162-164)
        runner = model.hip_runner()
        x = torch.randn(args.batch, 3, args.res, args.res, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        tgt = torch.randint(0, 1000, (args.batch,), device=dev, generator=g)
        from frostnet_amd.harness import CrossEntropyLoss
        crit = CrossEntropyLoss()                     # nn.CrossEntropyLoss semantics, forward + backward in one HIP kernel (harness.py)

        def loss_of(xx, tt):
            return crit(model(xx), tt)
        seg_kw = dict(criterion=crit)

    # N = 1: model(x) -> loss.backward() -> optimizer.step() (the reference loop, helper_functions.py:139-143) captured as ONE hipGraph.
    # N > 1 (default): the same kernels as a chain of hipGraph segments, one per gradient bucket; the bucket's RCCL all-reduce is issued
    # between segments and runs on RCCL's stream under the rest of the backward (frostnet_amd.parallel.SegmentedStep).
    seg = SegmentedStep(runner, nbuckets=args.buckets, **seg_kw) if (dp and not args.single_allreduce) else None

    # d(step loss)/d(loss): 1 on one GPU, 1 / world under data parallel (the mean over the global batch; gradients are linear in it, so the SUM all-reduce of the
    # shard gradients is the mean).  A persistent tensor handed to backward(): no root-gradient fill and no `loss / world` node in the captured step.
    gscale = torch.full((), 1.0 / (dist.get_world_size() if dp else 1), dtype=torch.float32, device=dev)

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)               # the reference loop (helper_functions.py:139); host-side only: p.grad = None, nothing is launched
        loss = loss_of(x, tgt)
        loss.backward(gradient=gscale)
        return loss

    ctl = {"comm": True}                          # the exposed-communication measurement replays the step with the collectives switched off

    def eager_step():
        if seg is not None:
            seg.run_eager(x, tgt)
            seg.finish()
        else:
            fwd_bwd()
            if dp and ctl["comm"]:
                dist.all_reduce(runner.grad_arena)
        opt.step()

    for _ in range(max(1, min(args.warmup, 3))):      # first steps build tables / state before any capture
        eager_step()
    torch.cuda.synchronize()
    det_bytes = None
    if detect:                                        # algorithmic bytes / image of the detector step, counted from the tape of one training forward (see side_workload)
        opt.zero_grad(set_to_none=True)
        l0 = loss_of(x, tgt)
        det_bytes = tape_algorithmic_bytes(runner.E.tape, x.numel()) // args.batch
        l0.backward(gradient=gscale)
        if dp:
            dist.all_reduce(runner.grad_arena)
        opt.step()
        del l0
        torch.cuda.synchronize()

    allreduce_check = None
    if dp and args.check_allreduce:
        # ONE backward pass on the live process group: every bucket is snapshotted at the moment the backward hands it to the exchange (this rank's own
        # gradient), then exchanged; afterwards the snapshots of all ranks are gathered and summed on the side.  The arena must hold that sum = the mean of
        # the ranks' own shard gradients (each was produced from loss / world) up to the summation order of the collective.  (Two separate passes would
        # differ by the order of the backward's fp32 atomics, 1e-3 ... 2e-2 at small batches: that noise is not what this check is about.)
        sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        rng0 = runner.rng_state()
        mine = torch.zeros_like(runner.grad_arena)
        if seg is not None:
            exchange = seg._reduce

            def snapshot_then_reduce(i):
                for lo, hi in seg.cuts[i][1]:
                    mine[lo:hi].copy_(runner.grad_arena[lo:hi])
                exchange(i)
            seg._reduce = snapshot_then_reduce
            seg.run_eager(x, tgt)
            seg.finish()
            seg._reduce = exchange
        else:
            fwd_bwd()
            mine.copy_(runner.grad_arena)
            dist.all_reduce(runner.grad_arena)
        torch.cuda.synchronize()
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        want = torch.stack(parts).sum(0)
        rel = float((runner.grad_arena - want).norm() / (want.norm() + 1e-30))
        own = float((mine * world - want).norm() / (want.norm() + 1e-30))            # how far a single rank's gradient is from the mean: the check is not vacuous
        model.load_state_dict(sd0)
        runner.set_rng_state(rng0)
        allreduce_check = dict(rel_err_vs_mean_of_rank_gradients=rel, single_rank_vs_mean=own, ok=bool(rel <= 1e-5), devices=_rank_devices(dev, world))
        if rank == 0:
            print(f"[bench] all-reduce check: arena vs sum of the ranks' own (loss / world) gradients {rel:.2e} (one rank alone: {own:.2e})", file=sys.stderr, flush=True)

    graph = None
    capture_fallback = None
    if not args.no_graph:
        try:
            if seg is not None:
                seg.capture(x, tgt)
                graph = seg.graphs
            else:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    plan = opt.prepare_step()
                    graph = torch.cuda.CUDAGraph(keep_graph=True)       # the hipGraph stays queryable (graph_census); instantiated below
                    # RCCL's watchdog thread polls events concurrently: restrict the capture check to this thread when a process group exists
                    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local" if dp else "global"):
                        fwd_bwd()
                        if not dp:
                            opt.launch(plan)
                    graph.instantiate()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
        except Exception as e:  # pragma: no cover
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
            graph = None
            torch.cuda.synchronize()
            if dp and not args.allow_fallback:
                # a degraded data-parallel run must never print a headline number: every rank leaves with the same code
                print(f"[bench] rank {rank}: the data-parallel step could not be captured; refusing to time a fallback (pass --allow-fallback to measure the degraded form)",
                      file=sys.stderr, flush=True)
                sys.exit(4)
            if seg is not None:
                # segment capture under a live RCCL watchdog failed: fall back to ONE graph + one post-backward all-reduce (the --single-allreduce form)
                seg.graphs = None
                seg = None
                capture_fallback = f"segmented capture failed ({type(e).__name__}); fell back to one graph + single all-reduce"
                try:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                            fwd_bwd()
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                except Exception as e2:
                    print(f"[bench] single-graph capture failed too ({type(e2).__name__}: {e2}); running eagerly", file=sys.stderr, flush=True)
                    graph = None
                    capture_fallback += "; that capture failed as well: eager launches"
                    torch.cuda.synchronize()

    def step():
        if graph is None:
            return eager_step()
        plan = opt.prepare_step()
        if seg is not None:
            seg.replay()
            seg.finish()
            opt.launch(plan)
        else:
            graph.replay()
            if dp:
                if ctl["comm"]:
                    dist.all_reduce(runner.grad_arena)
                opt.launch(plan)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt

    comm = None
    if dp:
        # (a) every bucket's all-reduce alone on an otherwise idle GPU (median of 5); (b) the EXPOSED part of the exchange = step time with the
        # collectives minus step time without them (the gradients of those extra steps stay un-reduced: they come after the timed region)
        buckets = seg.bucket_ranges() if seg is not None else [[(0, runner.grad_arena.numel())]]
        per_bucket = []
        for ranges in buckets:                  # a bucket = one contiguous arena range (the classifier) or a few (the detector's loc / conf heads): one collective per range
            ts = []
            for _ in range(5):
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for lo, hi in ranges:
                    dist.all_reduce(runner.grad_arena[lo:hi])
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            per_bucket.append(dict(bytes=4 * sum(hi - lo for lo, hi in ranges), ranges=len(ranges), allreduce_us=round(sorted(ts)[2] * 1e3, 1)))
        k = max(3, min(args.steps, 10))
        ctl["comm"] = False
        keep_reduce = seg._reduce if seg is not None else None
        if seg is not None:
            seg._reduce = lambda i: None
        step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        dt_nc = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dt_nc, op=dist.ReduceOp.MAX)
        ctl["comm"] = True
        if seg is not None:
            seg._reduce = keep_reduce
        ms_nc = float(dt_nc.item()) / k * 1e3
        total_ar = sum(b["allreduce_us"] for b in per_bucket) / 1e3
        comm = dict(mode=("single all-reduce after the backward" if seg is None else f"{len(seg.cuts)} buckets overlapped with the backward, one hipGraph segment per bucket"),
                    buckets=per_bucket, allreduce_ms_sum=round(total_ar, 3), step_ms_without_collectives=round(ms_nc, 3),
                    exposed_comm_ms_per_step=round(max(0.0, ms - ms_nc), 3), hidden_comm_ms_per_step=round(max(0.0, total_ar - max(0.0, ms - ms_nc)), 3),
                    fallback=capture_fallback, allreduce_check=allreduce_check)

    roofline = None
    if rank == 0 and not args.no_roofline and not detect:
        # per-kernel HIP-event timing on the launch stream (eager pass: events cannot be read out of a replayed graph)
        def local_step():                  # no collective in here: only rank 0 runs this leg
            crit(model(x), tgt).backward()
            opt.step()
        L.PROFILER = L.Profiler()
        local_step()
        summ = L.PROFILER.summary()
        dom = max(summ, key=lambda k: summ[k]["total_ms"])
        L.PROFILER = L.Profiler(only=dom)
        for _ in range(3):
            local_step()
        s2 = L.PROFILER.summary()[dom]
        L.PROFILER = None
        achieved = s2["bytes_per_launch"] / (s2["avg_ms"] * 1e-3) / 1e9
        pmc_batch = args.batch if args.res == 224 and args.mode == "large" else -1
        traffic, traffic_src = pmc_traffic(dom, pmc_batch)
        step_traffic, step_launches, step_src = pmc_step_traffic(pmc_batch)
        algo_step = ALGO_BYTES_PER_IMG * args.batch
        step_gbs = value / world * ALGO_BYTES_PER_IMG / 1e9
        # The headline roofline figure is the WHOLE STEP: algorithmic bytes of one image (layer-granular) x images / s over the
        # 8 TB/s HBM peak; `traffic` = HBM bytes of one step summed over every kernel family of the committed counter passes.  The dominant kernel family
        # (the contract's per-kernel figure: algorithmic bytes per launch / average launch duration by HIP events on the launch stream) is `dominant_kernel`.
        census = graph_census([graph]) if (graph is not None and seg is None and not isinstance(graph, list)) else None
        # ONE launch count in the line: this library's kernel nodes of the captured step, read back from the hipGraph of THIS run; the count of the
        # committed counter pass (an eager step under rocprofv3) only when no graph was captured
        own = census.get("kernels_own") if isinstance(census, dict) else None
        roofline = dict(bound="hbm", kernel="whole step", achieved=round(step_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(step_gbs / HBM_PEAK_GBS, 4), traffic=step_traffic, traffic_source=step_src,
                        traffic_total_bytes_per_step=step_traffic, algorithmic_bytes_per_step=algo_step,
                        traffic_ratio=(round(step_traffic / algo_step, 3) if step_traffic else None),
                        kernel_launches_per_step=(own if own else step_launches),
                        kernel_launches_source=("hipGraph nodes of this run (graph_nodes.kernels_own)" if own else step_src),
                        algorithmic_bytes_per_image=ALGO_BYTES_PER_IMG,
                        # the captured step itself, node by node (read back from the hipGraph): own kernels / aten kernels / memset / memcpy
                        graph_nodes=census,
                        dominant_kernel=dict(kernel=dom, achieved=round(achieved, 1), unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                                             traffic=traffic, traffic_source=traffic_src, avg_launch_ms=round(s2["avg_ms"], 4),
                                             launches_per_step=s2["launches"] // 3, algorithmic_bytes_per_launch=int(s2["bytes_per_launch"]),
                                             share_of_step=round(summ[dom]["total_ms"] / max(1e-9, sum(v["total_ms"] for v in summ.values())), 3)),
                        # secondary roof (north_star's "MFMA utilisation"): fwd convs on the int8 MFMA (861.6 MFLOP/img), dgrad + wgrad on
                        # the bf16 MFMA (1701.6 MFLOP/img); dense peaks 5 POP/s int8, 2.5 PFLOP/s bf16.  The path is HBM-bound.
                        mfma=dict(achieved=round(value / world * FLOPS_PER_IMG / 1e12, 2), unit="TFLOP/s",
                                  frac_of_time_at_peak=round(value / world * (FLOPS_I8_PER_IMG / 5.0e15 + FLOPS_BF16_PER_IMG / 2.5e15), 4),
                                  flops_per_image=FLOPS_PER_IMG),
                        breakdown_ms={k: round(v["total_ms"], 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])},
                        # every family: launches per step, average launch (us, HIP events), algorithmic MB per launch, achieved GB/s, and the
                        # PMC traffic ratio (HBM bytes from the committed counter passes / algorithmic bytes)
                        families={k: dict(n=v["launches"], us=round(v["avg_ms"] * 1e3, 1), mb=round(v["bytes_per_launch"] / 1e6, 1),
                                          gbs=round(v["bytes_per_launch"] / (v["avg_ms"] * 1e-3) / 1e9, 0),
                                          traffic_ratio=(round(pmc_traffic(k, args.batch)[0] / v["bytes_per_launch"], 2)
                                                         if pmc_traffic(k, args.batch)[0] and args.res == 224 and args.mode == "large" else None))
                                  for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])})
    if rank == 0 and detect and det_bytes:
        gbs = value / world * det_bytes / 1e9
        roofline = dict(bound="hbm", kernel="whole step", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None,
                        algorithmic_bytes_per_image=det_bytes, algorithmic_bytes_per_step=det_bytes * args.batch)
    if world > 1:
        dist.barrier()

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1 and not detect:
            try:
                cpu = cpu_baseline()
            except Exception as e:  # pragma: no cover
                cpu = dict(value=None, error=str(e))
        extras = {}
        if world == 1 and not dp and not detect and not args.no_extras and graph is not None:
            # (1) the mode that meets north_star's 1e-3 gradient statement (FROST_GRAD=fp32: fp32 activation gradients, fp64 long sums) gets a driver-timed number
            # beside the bf16-gradient headline; (2) the N > 1 code path -- chain of hipGraph segments with the bucket's RCCL all-reduce between them -- on a 1-rank
            # RCCL group: its cost over the single graph is the segmentation + collective-launch overhead every rank pays at N = 8
            shape = ["--batch", str(args.batch), "--res", str(args.res), "--mode", args.mode]
            g32 = extra_leg(shape + ["--steps", "5", "--warmup", "2"], {"FROST_GRAD": "fp32"})
            extras["fp32_grad_ms_per_step"] = g32.get("ms_per_step")
            if "error" in g32:
                extras["fp32_grad_error"] = g32["error"]
            dpl = extra_leg(shape + ["--steps", "10", "--warmup", "3", "--force-dp"], {"FROST_RDZV_TIMEOUT": "60"})          # (a communicator bring-up that stalls costs a minute, not the line)
            if "error" in dpl:
                extras["dp_overhead_ms"] = None
                extras["dp_overhead_error"] = dpl["error"]
            else:
                gar = (dpl.get("config") or {}).get("grad_allreduce") or {}
                extras["dp_overhead_ms"] = round(dpl["ms_per_step"] - ms, 3)
                extras["dp_overhead"] = dict(what="bench.py --force-dp: 1-rank RCCL group, chain of hipGraph segments + one all-reduce per bucket, minus this run's single-graph step",
                                             segmented_ms_per_step=dpl["ms_per_step"], single_graph_ms_per_step=round(ms, 3), mode=gar.get("mode"),
                                             bucket_bytes=[b.get("bytes") for b in gar.get("buckets", [])],
                                             bucket_allreduce_us_1rank=[b.get("allreduce_us") for b in gar.get("buckets", [])],
                                             step_ms_without_collectives=gar.get("step_ms_without_collectives"))
        metric = (f"images/sec SSDLite-FrostNet-{args.mode.capitalize()} {args.res}x{args.res} QAT fwd+bwd" if detect
                  else "images/sec FrostNet-Large 224x224 QAT fwd+bwd")
        what = (f"SSDLite on the FrostNet-{args.mode.capitalize()} backbone (frostnet_amd.ssdlite), int8 fake-quant QAT fwd+bwd + MultiBoxLoss + GradBoost-SGD step, "
                f"batch={args.batch}/GPU, {args.res}x{args.res} (BASELINE.json config c5, data parallel)" if detect else
                f"FrostNet-{args.mode.capitalize()} int8 fake-quant QAT fwd+bwd + GradBoost-SGD step "
                f"(noise on; StatAssist FP epoch is a one-off before it), batch={args.batch}/GPU, "
                f"{args.res}x{args.res} NHWC, qnnpack qconfig v0 (per-tensor)")
        out = dict(metric=metric, value=round(value, 2), unit="images/sec",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="int8", data="synthetic",
                   config=dict(workload=what,
                               per_gpu_batch=args.batch, global_batch=args.batch * world, resolution=args.res,
                               parallelism=f"dp{world}", hip_graph=graph is not None, grad_dtype=("fp32" if runner.E.grad_fp32 else ("mixed (fp32 on maps <= 14 x 14, bf16 above)" if getattr(runner.E, "grad_mixed", False) else "bf16")),
                               ms_per_step_median_hip_events=round(ms_median, 3),
                               grad_allreduce=comm, **extras),
                   roofline=roofline, cpu_baseline=cpu)
    if dp:
        dist.destroy_process_group()
    if rank == 0:
        try:                                      # RCCL writes its banner through C stdio: flush it so the JSON line stays last
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # pragma: no cover
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
