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
from torch import nn


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


# ---------------------------------------------------------------------------------------------------------------------
# Data-parallel training (BASELINE.json configs[4]): gradient all-reduce overlapped with backward, SyncBatchNorm
# ---------------------------------------------------------------------------------------------------------------------
class GradientSynchronizer:
    """Bucketed gradient all-reduce that runs WHILE backward is still producing gradients.

    The parameters' `.grad` tensors are views into a few flat, persistent bucket buffers (no per-step cat / copy
    back).  Buckets are filled in reverse parameter order - the order backward reaches them - and a
    post-accumulate-grad hook on every parameter counts its bucket down; when the last gradient of a bucket has been
    written its all-reduce(SUM) is launched asynchronously, so the collectives of the last layers' buckets run on
    RCCL's stream under the backward GEMMs of the earlier layers.  `finish()` waits for the collectives and scales by
    1 / world.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few large buckets (64 MB default; the cfg-3
    model has 214 MB of fp32 gradients) keep every ring busy without serialising behind one giant message.

        sync = GradientSynchronizer(model.parameters())
        for batch in shard:
            sync.zero_grad()                  # zeroes the flat buffers in place (the views stay)
            loss(batch).backward()            # all-reduces start as buckets complete
            sync.finish()                     # gradients are now the mean over ranks
            optimizer.step()

    One backward() per zero_grad() / finish() pair: a gradient that arrives for a bucket whose all-reduce is already
    under way raises (it would race with the collective); accumulate with `with sync.no_sync(): loss.backward()` for
    all but the last micro-batch.

    The reference has no counterpart (single process).  With world size 1 (or no process group) it only provides
    the flat views and `finish()` is a no-op."""

    def __init__(self, parameters, group=None, bucket_bytes=64 << 20, average=True):
        self.group, self.average = group, average
        self.params = [p for p in parameters if p.requires_grad]
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.buckets = []                 # [flat, [params], pending, work]
        self._bucket_of = {}
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self._make_bucket(cur)
                cur, size = [], 0
        if cur:
            self._make_bucket(cur)
        self._defer = False
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        # the package's own autograd nodes accumulate weight gradients straight into these views (autograd._grad_sink)
        # and call the hooks themselves: use loss.backward(), not torch.autograd.grad, on these parameters meanwhile
        from . import autograd as _ag
        _ag._engine_fires_post_accumulate_hooks_without_a_gradient()      # (probed here, outside any backward pass)
        for p in self.params:
            p._nf_grad_sink = True

    def no_sync(self):
        """Context manager for gradient accumulation: backward passes inside it only add into the flat buffers; the
        buckets are reduced by the next backward outside it (hooks) or by finish()."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev, self._defer = self._defer, True
            try:
                yield self
            finally:
                self._defer = prev
        return cm()

    def _make_bucket(self, params):
        dtype, device = params[0].dtype, params[0].device
        if any(p.dtype != dtype or p.device != device for p in params):
            raise ValueError('parameters of one bucket must share dtype and device')
        flat = torch.zeros(sum(p.numel() for p in params), dtype=dtype, device=device)
        off, views = 0, []
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        b = {'flat': flat, 'params': list(params), 'views': views, 'pending': len(params), 'work': None}
        for p, v in zip(params, views):
            self._bucket_of[p] = (b, v)
            p.grad = v
        self.buckets.append(b)

    def zero_grad(self):
        for b in self.buckets:
            b['flat'].zero_()
            b['pending'], b['work'] = len(b['params']), None
            for p, v in zip(b['params'], b['views']):
                if p.grad is not v:
                    p.grad = v

    def _hook(self, p):
        b, v = self._bucket_of[p]
        if p.grad is not v:                   # something replaced the view (zero_grad(set_to_none=True) ...): restore it
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v
        if b['pending'] <= 0 or b['work'] is not None:
            # a second backward() before finish(): the bucket's all-reduce may still be reading the flat buffer this
            # gradient was just accumulated into, and the count can no longer tell when the bucket is complete
            raise RuntimeError('GradientSynchronizer: a gradient arrived for a bucket that is already complete - '
                               'call finish() (and zero_grad()) between backward passes; to accumulate gradients over '
                               'several backward passes wrap all but the last in `with sync.no_sync():`')
        if self._defer:
            return
        b['pending'] -= 1
        if b['pending'] == 0 and self.world > 1:
            b['work'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Waits for the bucket collectives (buckets whose hooks did not all fire - unused parameters - are reduced
        here) and averages.  Returns the number of buckets that were reduced while backward was still running."""
        early = 0
        for b in self.buckets:
            if self.world > 1:
                if b['work'] is None:
                    b['work'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                else:
                    early += 1
        for b in self.buckets:
            if b['work'] is not None:
                b['work'].wait()
                if self.average:
                    b['flat'].div_(self.world)
            b['pending'], b['work'] = len(b['params']), None
        return early

    def remove(self):
        for h in self._handles:
            h.remove()
        for p in self.params:
            p._nf_grad_sink = False


class _SyncBatchNormFn(torch.autograd.Function):
    """Training-mode batch normalisation over the GLOBAL batch.  Forward: every rank computes [n, mean, M2 = sum
    (x - mean)^2] of its shard (the centred form nn.BatchNorm1d itself uses: no cancellation when |mean| >> std), ONE
    all-gather of those 2F+1 float64 values, and the pairwise (Chan et al.) combination on every rank in float64 and in
    rank order - so all ranks hold bit-identical statistics.  Backward: one all-reduce(SUM) of [sum dy, sum dy*xhat].
    weight / bias may be None (affine=False)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        F_ = x.shape[1]
        n_local = x.shape[0]
        x64 = x.double()
        if n_local:
            mean_l = x64.mean(0)
            m2_l = ((x64 - mean_l) ** 2).sum(0)
        else:                                   # an empty shard still takes part in the collective
            mean_l = x64.new_zeros(F_)
            m2_l = x64.new_zeros(F_)
        part = torch.cat([x64.new_full((1,), float(n_local)), mean_l, m2_l])
        world = dist.get_world_size(group)
        parts = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(parts, part, group=group)
        allp = torch.stack(parts)                                   # [world, 2F+1]
        ns = allp[:, :1]
        n = ns.sum()
        mean64 = (ns * allp[:, 1:1 + F_]).sum(0) / n
        m2 = (allp[:, 1 + F_:] + ns * (allp[:, 1:1 + F_] - mean64) ** 2).sum(0)
        var64 = m2 / n                                              # biased, as F.batch_norm normalises with
        invstd = torch.rsqrt(var64 + eps).to(x.dtype)
        mean = mean64.to(x.dtype)
        xhat = (x - mean) * invstd
        ctx.save_for_backward(xhat, weight, invstd)
        ctx.group, ctx.n, ctx.has_bias = group, n.to(x.dtype), bias is not None
        var = var64.to(x.dtype)
        n_out = n.to(x.dtype)
        ctx.mark_non_differentiable(mean, var, n_out)
        y = xhat if weight is None else xhat * weight
        if bias is not None:
            y = y + bias
        return y, mean, var, n_out

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gn):
        xhat, weight, invstd = ctx.saved_tensors
        F_ = xhat.shape[1]
        sums = torch.cat([gy.sum(0), (gy * xhat).sum(0)])
        gb_local, gw_local = sums[:F_].clone(), sums[F_:].clone()         # parameter gradients stay LOCAL sums: the
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=ctx.group)      # gradient all-reduce adds them up like any other
        n = ctx.n
        gx = (gy - sums[:F_] / n - xhat * (sums[F_:] / n)) * (invstd if weight is None else weight * invstd)
        return gx, (None if weight is None else gw_local), (gb_local if ctx.has_bias else None), None, None


class SyncBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d whose train()-mode statistics span every rank's shard (reference: plain BatchNorm1d over the one
    process's whole batch, made.py:232-236 / resnet.py:22-26 - sharding the batch must not change what it computes).
    Same parameters, buffers and state_dict keys; eval() mode and single-process runs are nn.BatchNorm1d itself."""

    group = None

    def forward(self, x):
        if not (self.training and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return super().forward(x)
        if x.dim() != 2:
            raise ValueError('expected 2D input (got {}D input)'.format(x.dim()))
        y, mean, var, n = _SyncBatchNormFn.apply(x, self.weight, self.bias, self.eps, self.group)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                unbiased = var * (n / (n - 1).clamp_min(1.0))
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(unbiased, alpha=m)
        return y


def convert_sync_batchnorm(module, group=None):
    """Replaces every nn.BatchNorm1d inside `module` (MADE / ResidualNet blocks built with use_batch_norm=True) by a
    SyncBatchNorm1d sharing its parameters and buffers; returns `module`."""
    for name, child in list(module.named_children()):
        if isinstance(child, nn.BatchNorm1d) and not isinstance(child, SyncBatchNorm1d):
            sbn = SyncBatchNorm1d(child.num_features, eps=child.eps, momentum=child.momentum, affine=child.affine,
                                  track_running_stats=child.track_running_stats)
            sbn.group = group
            if child.affine:
                sbn.weight, sbn.bias = child.weight, child.bias
            if child.track_running_stats:
                sbn.running_mean, sbn.running_var = child.running_mean, child.running_var
                sbn.num_batches_tracked = child.num_batches_tracked
            sbn.train(child.training)
            if isinstance(module, nn.ModuleList):
                module[int(name)] = sbn
            else:
                setattr(module, name, sbn)
        else:
            convert_sync_batchnorm(child, group)
    return module
