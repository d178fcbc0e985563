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


def _rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def streamed_log_likelihood(log_prob_fn, n_global, make_rows, rows_per_pass, group=None):
    """BASELINE.json configs[3] (batch far larger than memory, e.g. 64 M x 784 = 200 GB of inputs): this rank's
    contiguous shard [lo, hi) of `n_global` rows is produced `rows_per_pass` rows at a time by `make_rows(s, e)` (a
    device-side generator, or a slice of a memory-mapped file: nothing of size [shard, D] is ever resident), pushed
    through `log_prob_fn`, and only an fp64 [sum log p, count] pair survives a pass.  One all-reduce(SUM) of that pair
    at the end is the whole exchange.  The last pass of a shard may be short; shards differ by at most one row.
    Returns (sum over all ranks, count over all ranks, this rank's (lo, hi))."""
    if rows_per_pass <= 0:
        raise ValueError('rows_per_pass must be positive')
    rank, world = _rank_world(group)
    lo, hi = shard_bounds(n_global, rank, world)
    acc = None
    for s in range(lo, hi, rows_per_pass):
        e = min(hi, s + rows_per_pass)
        lp = log_prob_fn(make_rows(s, e))
        if lp.shape[0] != e - s:
            raise ValueError('log_prob_fn returned %d values for %d rows' % (lp.shape[0], e - s))
        if acc is None:
            acc = torch.zeros(2, dtype=torch.float64, device=lp.device)
        acc[0] += lp.double().sum()
        acc[1] += e - s
    if acc is None:                                     # an empty shard still takes part in the reduction
        acc = torch.zeros(2, dtype=torch.float64, device=_reduce_device(group))
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    total, count = acc.tolist()
    return total, count, (lo, hi)


def streamed_sampling(sample_fn, n_global, rows_per_pass, group=None):
    """The sampling half of configs[3]: this rank draws its shard of `n_global` samples `rows_per_pass` at a time with
    `sample_fn(n) -> (x [n, D], log q [n])` (distributions.sample_and_log_prob); samples stay on the rank that drew
    them (or are consumed there) and only fp64 moments cross ranks: one all-reduce(SUM) of
    [count, sum log q, sum_b x (D values), sum_b x^2 (D values)].
    Returns dict(count, mean_logq, mean [D], var [D]) over all ranks."""
    if rows_per_pass <= 0:
        raise ValueError('rows_per_pass must be positive')
    rank, world = _rank_world(group)
    lo, hi = shard_bounds(n_global, rank, world)
    acc = None
    for s in range(lo, hi, rows_per_pass):
        n = min(hi, s + rows_per_pass) - s
        x, logq = sample_fn(n)
        x64 = x.double().reshape(n, -1)
        part = torch.cat([torch.tensor([float(n)], dtype=torch.float64, device=x.device), logq.double().sum().reshape(1),
                          x64.sum(0), (x64 * x64).sum(0)])
        acc = part if acc is None else acc + part
    if world > 1:
        # every rank must agree on the vector's length: ranks with an empty shard learn it from the others
        size = torch.tensor([0 if acc is None else acc.numel()], dtype=torch.int64, device=_reduce_device(group))
        dist.all_reduce(size, op=dist.ReduceOp.MAX, group=group)
        if acc is None:
            acc = torch.zeros(int(size.item()), dtype=torch.float64, device=size.device)
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    if acc is None or acc[0].item() == 0:
        return {'count': 0.0, 'mean_logq': float('nan'), 'mean': None, 'var': None}
    cnt = acc[0].item()
    D = (acc.numel() - 2) // 2
    mean = acc[2:2 + D] / cnt
    return {'count': cnt, 'mean_logq': acc[1].item() / cnt, 'mean': mean, 'var': acc[2 + D:] / cnt - mean * mean}


def _reduce_device(group=None):
    """Device collectives of this group run on: the current GPU under RCCL, the CPU under gloo."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


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
