"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The hot path is row-wise independent over the batch, so the body needs no
collective: every rank evaluates its contiguous shard with replicated weights.  The only exchange
is the log-likelihood reduction at the end: one all-reduce(SUM) of an fp64 [sum log p, count] pair.
The reference has no counterpart (single process); this is what `.mean()` of model.py:133 becomes
when the batch is sharded.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n rows for `rank`; sizes differ by at most one row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_log_likelihood(log_prob, group=None):
    """(sum of log_prob over all ranks, number of samples over all ranks) as Python floats.
    Accumulated in float64 so the result does not depend on how the batch was sharded beyond
    rounding of the per-rank partial sums."""
    acc = torch.stack([log_prob.double().sum(), torch.tensor(float(log_prob.numel()), dtype=torch.float64,
                                                            device=log_prob.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    s, n = acc.tolist()
    return s, n


def sharded_log_prob(log_prob_fn, x_global, group=None):
    """Evaluates `log_prob_fn` on this rank's shard of `x_global`; returns (local log_prob, lo, hi,
    global mean log-likelihood)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(x_global.shape[0], rank, world)
    lp = log_prob_fn(x_global[lo:hi])
    s, n = reduce_log_likelihood(lp, group)
    return lp, lo, hi, (s / n if n else float('nan'))


def allreduce_gradients(parameters, group=None, bucket_bytes=64 << 20, average=True):
    """Data-parallel gradient synchronisation: gradients are flattened into buckets of ~`bucket_bytes`
    and each bucket is all-reduced (SUM) asynchronously, then scaled by 1/world and copied back.
    On one node xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a few large buckets per step
    (the cfg-3 model has 214 MB of fp32 gradients) keep every RCCL ring busy; the collectives of later
    buckets overlap the flatten/copy work of earlier ones.  The reference has no counterpart (single
    process).  Returns the number of buckets."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    grads = [p.grad for p in parameters if p.grad is not None]
    buckets, cur, size = [], [], 0
    for g in grads:
        cur.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    for work, flat, b in pending:
        work.wait()
        if average:
            flat.div_(world)
        off = 0
        for g in b:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
    return len(buckets)
