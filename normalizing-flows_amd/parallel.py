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
