"""What consumes `log_prob` in the reference's harness: effective sample size and the KL / JS estimates
(reference helper_functions.py:36-40 `compute_ess`; models/model.py:110-145
`compute_ess_from_distributions`, `compute_forward_kl_divergence`, `compute_reverse_kl_divergence`,
`compute_jensen_shanon_divergence`), on a batch that is sharded over the ranks of a process group.

Every quantity is a reduction of per-sample log-weights, so the sharded form needs no data movement: the
log-sum-exps become one all-reduce(MAX) of the local maximum followed by one all-reduce(SUM) of a few
float64 partial sums (RCCL on the GPUs, gloo in the CPU tests); the means one all-reduce(SUM) of
[sum, count].  With no process group the same code is the single-process formula."""
import math

import torch
import torch.distributed as dist


def _world(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _allreduce(t, op, group):
    if _world(group) > 1:
        dist.all_reduce(t, op=op, group=group)
    return t


def distributed_logsumexp(values, powers=(1,), group=None):
    """[log sum_i exp(p * v_i) for p in powers] over the shards of every rank, as Python floats.
    Returns also the global number of samples."""
    v = values.detach().double().reshape(-1)
    local_max = v.max() if v.numel() else torch.tensor(-math.inf, dtype=torch.float64, device=v.device)
    m = _allreduce(local_max.clone().reshape(1), dist.ReduceOp.MAX, group)[0]
    shift = m if torch.isfinite(m) else torch.zeros_like(m)
    sums = torch.stack([torch.exp(p * (v - shift)).sum() for p in powers]
                       + [torch.tensor(float(v.numel()), dtype=torch.float64, device=v.device)])
    sums = _allreduce(sums, dist.ReduceOp.SUM, group)
    out = [float(torch.log(sums[i]) + p * shift) for i, p in enumerate(powers)]
    return out, int(sums[-1].item())


def compute_ess(logp, logq, group=None):
    """Effective sample size per configuration, exp(2 lse(logw) - lse(2 logw)) / N with logw = logp - logq
    (reference helper_functions.py:36-40), N and both log-sum-exps taken over all ranks."""
    (l1, l2), n = distributed_logsumexp(logp - logq, powers=(1, 2), group=group)
    return math.exp(2.0 * l1 - l2) / n if n else float('nan')


def _global_mean(values, group):
    v = values.detach().double().reshape(-1)
    acc = torch.stack([v.sum(), torch.tensor(float(v.numel()), dtype=torch.float64, device=v.device)])
    acc = _allreduce(acc, dist.ReduceOp.SUM, group)
    return float(acc[0] / acc[1]) if acc[1] > 0 else float('nan')


def forward_kl(logp, logq, group=None):
    """mean(logp - logq) over samples of the target (reference model.py:126-130)."""
    return _global_mean(logp - logq, group)


def reverse_kl(logq, logp, group=None):
    """mean(logq - logp) over samples of the flow (reference model.py:132-136)."""
    return _global_mean(logq - logp, group)


def jensen_shannon(forward, reverse):
    """The reference's 0.5 (forward KL + reverse KL) (model.py:138-142)."""
    return 0.5 * (forward + reverse)


def flow_metrics(flow, target, batch_size, group=None):
    """ESS and the KL / JS estimates of `flow` (a BaseNeuralSplineFlow) against `target` (an object with
    `sample_n(n)` and `log_prob(x)`, like the reference's distributions): each rank draws `batch_size`
    samples of its own; reductions are global.  Mirrors model.py:110-142."""
    with torch.no_grad():
        x, logq = flow.sample_and_log_prob(batch_size)                  # apply_flow_to_prior
        logp = target.log_prob(x)
        y = target.sample_n(batch_size)                                 # apply_reverse_to_samples
        logq_y = flow.log_prob(y)
        logp_y = target.log_prob(y)
    fkl = forward_kl(logp_y, logq_y, group)
    rkl = reverse_kl(logq, logp, group)
    return {'ess': compute_ess(logp, logq, group), 'target_ess': compute_ess(logp_y, logq_y, group),
            'forward_kl': fkl, 'reverse_kl': rkl, 'jensen_shannon': jensen_shannon(fkl, rkl)}
