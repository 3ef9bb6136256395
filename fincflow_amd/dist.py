"""Batch sharding of the hot path over the GPUs of one node.

Every image (and each of its 4 groups) is an independent triangular system, so ranks take contiguous batch
slices and never exchange activations.  The only collective is one broadcast of the layer's weights per weight
version (<= 83 KB at C=96 3x3; RCCL over xGMI when the backend is "nccl", gloo in the CPU tests).  The reference
has no multi-GPU inverse to mirror: nn.DataParallel wraps the training forward only and sampling runs on
model.module on one device (train/experiment.py:311-314,328-332).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, balanced slice [lo, hi) of a batch of n for `rank` of `world` (first n % world ranks get one more)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t, rank=None, world=None):
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def broadcast_weights(module, src=0):
    """Replicate every parameter of `module` from rank `src` (in place) and drop the layers' canonical /
    packed-fragment caches (c10d collectives write through raw pointers and do not bump Tensor._version)."""
    with torch.no_grad():
        for prm in module.parameters():
            dist.broadcast(prm, src=src)
    for m in module.modules():
        cache = getattr(m, "_cache", None)
        if cache is not None and hasattr(cache, "invalidate"):
            cache.invalidate()
    return module


def max_over_ranks(seconds, device=None):
    """The bench's timing rule: a step is done when the slowest rank is done."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_shards(t, total):
    """Optional: reassemble a batch-sharded result on every rank.  Not part of the timed path (outputs stay
    sharded for the next per-image layer); shards may differ in size by one."""
    world = dist.get_world_size()
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)
