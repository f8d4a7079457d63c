"""Data-parallel QAT: one process per GPU, gradient-arena all-reduce over RCCL/xGMI overlapped with backward.

replaces: nn.DataParallel's replicate/scatter/gather/ReduceAddCoalesced (Classification/train.py:88-92) and timm's DDP
(training_commands.txt) -- SURVEY.md 2.3 C1-C5.  Each rank owns its replica, its data shard and its own BN / observer
state (never synchronised in the reference either); the ONE exchange step is the sum of the 5.8 M-element fp32
gradient arena.  Backward produces gradients in reverse parameter order, so the arena is cut into a few contiguous
buckets (xGMI is point-to-point, ring collectives are per-link bound: few large messages, not many small ones) and
a bucket's all-reduce is launched asynchronously the moment the backward pass has finished every layer inside it.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, arena, param_offsets, nbuckets=4, group=None):
        """arena: flat fp32 gradient buffer; param_offsets: start offset of every parameter (ascending)."""
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = arena.numel()
        offs = sorted(set(param_offsets))
        bounds = [0]
        for b in range(1, nbuckets):
            target = n * b // nbuckets
            cand = min(offs, key=lambda o: abs(o - target))
            if cand > bounds[-1]:
                bounds.append(cand)
        bounds.append(n)
        self.buckets = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
        self.reset()

    def reset(self):
        self._launched = [False] * len(self.buckets)
        self._handles = []

    def ready(self, offset):
        """Every gradient at arena offset >= `offset` is final: launch the buckets that are now complete."""
        if self.world == 1:
            return
        for i in reversed(range(len(self.buckets))):
            a, b = self.buckets[i]
            if not self._launched[i] and a >= offset:
                self._handles.append(dist.all_reduce(self.arena[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self._launched[i] = True

    def finish(self):
        """Launch whatever is left, wait, and turn the sum into the mean (loss is a local mean per rank)."""
        if self.world == 1:
            return
        self.ready(0)
        for h in self._handles:
            h.wait()
        self.arena.mul_(1.0 / self.world)
        self.reset()


def broadcast_model(model, src=0, group=None):
    """One-time parameter/buffer broadcast at start (replaces DataParallel's per-forward broadcast_coalesced)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    # coalesced like DataParallel's broadcast_coalesced (Classification/train.py:88-92): one flat buffer per (dtype, device) instead of ~1.6 k collectives of a few
    # hundred bytes each (FrostNet-Large: 212 parameters + ~1.4 k observer / BatchNorm buffers)
    groups = {}
    for t in list(model.parameters()) + list(model.buffers()):
        groups.setdefault((t.dtype, t.device), []).append(t.data)
    for (dtype, _dev), ts in groups.items():
        wire = torch.uint8 if dtype == torch.bool else dtype
        flat = torch.cat([t.reshape(-1).to(wire) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off: off + n].view(t.shape).to(dtype))
            off += n


def plan_buckets(runner, nbuckets=4):
    """Cut the gradient arena into `nbuckets` buckets along LAYER boundaries, in the order the backward pass completes them (for the
    classifier: classifier first, stem last; for the SSDLite detector: prediction heads, extras, then the backbone from layer5 down).
    Returns [(layer, ranges)]: once `layer` has run its backward every gradient in the arena ranges [(lo, hi), ...] is final.  A bucket
    is ONE contiguous range wherever the parameter order follows the layer order (the whole classification network; the backbone and
    the extras of the detector) and a few ranges where it does not (the detector registers `loc` and `conf` as two ModuleLists while
    the backward visits loc_i / conf_i interleaved).  FrostNet's parameters sit at the deep end (classifier 1.28 M + last_layer 0.41 M +
    layer5 1.1 M of 5.8 M), its backward *time* at the shallow end, so the first buckets carry most of the bytes and travel under almost
    the whole backward pass; the last bucket is a few hundred KB."""
    offs, off = {}, 0
    for p in runner._params:
        offs[p.data_ptr()] = (off, off + p.numel())
        off += p.numel()
    total = off
    layers = [l for l in runner.E.layers]                    # forward order = reverse of the order the backward finishes them
    cuts, pend, done, k = [], [], 0, 1
    for l in reversed(layers):
        for t in (l.w, l.gamma, l.beta, l.bias):
            if t is not None:
                pend.append(offs[t.data_ptr()])
                done += t.numel()
        if l is layers[0] or done >= total * k / nbuckets:
            cuts.append((l, _coalesce(pend)))
            pend = []
            while done >= total * k / nbuckets:
                k += 1
    covered = _coalesce([r for _, rs in cuts for r in rs])
    assert covered == [(0, total)], f"gradient buckets do not tile the arena: {covered} vs {total}"
    return cuts


def _coalesce(ranges):
    out = []
    for lo, hi in sorted(ranges):
        if out and out[-1][1] == lo:
            out[-1] = (out[-1][0], hi)
        else:
            out.append((lo, hi))
    return out


class SegmentedStep:
    """One rank's forward + backward as a CHAIN of hipGraph segments, one per gradient bucket, with the bucket's RCCL
    all-reduce issued between segments: segment k+1 (the rest of the backward) replays on the compute stream while bucket k is
    being reduced over xGMI on RCCL's stream.  The collective itself is never captured (plain torch.distributed launch), so
    the N-GPU path needs nothing from RCCL beyond what eager training uses.

    replaces: nn.DataParallel's gather / ReduceAddCoalesced (Classification/train.py:88-92, Object_Detection/qtrainval.py:123-127), timm DDP's
    bucketed overlap (training_commands.txt).  The gradient of the step loss w.r.t. the local loss is 1 / world, so the SUM all-reduce
    yields the mean gradient with no extra pass over the arena (exact: gradients are linear in it).

    Two step bodies: `criterion` (classification: logits -> criterion(logits, target)) or `maps_loss` (a model whose runner returns several
    fake-quantised maps -- SSDRunner._maps_impl: maps_loss(list of dequantised fp32 maps, target) -> scalar loss; BASELINE.json config c5).
    Either way torch autograd only differentiates the loss; the hand-written backward is called on THIS thread (segment captures must begin
    and end on the thread that started them)."""

    def __init__(self, runner, criterion=None, nbuckets=4, group=None, maps_loss=None):
        assert (criterion is None) != (maps_loss is None), "SegmentedStep: give either a classification criterion or a maps_loss"
        self.runner, self.crit, self.maps_loss, self.group = runner, criterion, maps_loss, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuts = plan_buckets(runner, nbuckets)
        self.boundaries = {id(l): i for i, (l, _) in enumerate(self.cuts)}
        self.graphs, self._handles, self.loss = None, [], None
        self.gscale = torch.full((), 1.0 / self.world, dtype=torch.float32, device=runner.device)

    def bucket_ranges(self):
        """[(lo, hi), ...] per bucket, in the order the backward completes them."""
        return [list(rs) for _, rs in self.cuts]

    # -- the step body: forward, loss, gradient of the loss w.r.t. the network outputs (torch autograd only for the loss), hand-written backward
    def _body(self, x, target, on_bucket):
        r = self.runner
        if self.maps_loss is not None:
            from .engine import float_to_grad
            acts = r._maps_impl(x, record=True)
            leaves = [a.dequant().requires_grad_(True) for a in acts]
            loss = self.maps_loss(leaves, target)
            loss.backward(gradient=self.gscale)
            for a, t in zip(acts, leaves):
                g = t.grad if t.grad is not None else torch.zeros_like(t)
                a.grad = float_to_grad(g, fp32=r.E.grad_is_fp32(a))
            r.bind_grads()
            r.E.backward(None, boundaries=self.boundaries, on_bucket=on_bucket)
            return loss.detach()
        logits = r._forward_impl(x, record=True).detach().requires_grad_(True)
        loss = self.crit(logits, target)
        loss.backward(gradient=self.gscale)
        r.bind_grads()
        r.E.backward(logits.grad, boundaries=self.boundaries, on_bucket=on_bucket)
        return loss.detach()

    def _reduce(self, i):
        for lo, hi in self.cuts[i][1]:
            self._handles.append(dist.all_reduce(self.runner.grad_arena[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for h in self._handles:
            h.wait()
        self._handles = []

    def run_eager(self, x, target):
        """Eager launches, bucket all-reduces started from inside the backward pass."""
        self.loss = self._body(x, target, self._reduce if dist.is_initialized() else None)
        return self.loss

    def capture(self, x, target):
        """Capture the step on static inputs as len(cuts) graphs sharing one memory pool.  Run a few eager steps first: descriptor
        tables and state are built on first use and must not be created during capture."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graphs, state = [], {}
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            g.capture_begin(capture_error_mode="thread_local")
            state["g"] = g

            def on_bucket(i):
                state["g"].capture_end()
                graphs.append(state["g"])
                if i + 1 < len(self.cuts):
                    state["g"] = torch.cuda.CUDAGraph()
                    state["g"].capture_begin(pool=graphs[0].pool(), capture_error_mode="thread_local")
            self.loss = self._body(x, target, on_bucket)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert len(graphs) == len(self.cuts)
        self.graphs = graphs
        return self.loss

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if dist.is_initialized():
                self._reduce(i)
        return self.loss
