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
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
