"""ESS / KL / JS reductions (reference helper_functions.py:36-40, model.py:110-142) and the checkpoint
format: single process against the oracle's restatement, and two gloo ranks against one process."""
import math
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)


def _data(n, seed=5):
    g = torch.Generator().manual_seed(seed)
    logp = torch.randn(n, generator=g) * 3.0 - 40.0
    logq = logp + torch.randn(n, generator=g) * 0.7 + 0.3
    return logp, logq


@pytest.mark.parametrize('n', [1, 17, 4096])
def test_metrics_match_the_oracle_single_process(n):
    from normalizing_flows_amd import metrics
    from oracle import nf_oracle as O
    logp, logq = _data(n)
    assert abs(metrics.compute_ess(logp, logq) - O.compute_ess(logp.double(), logq.double()).item()) < 1e-9
    assert abs(metrics.forward_kl(logp, logq) - O.forward_kl(logp.double(), logq.double()).item()) < 1e-9
    assert abs(metrics.reverse_kl(logq, logp) - O.reverse_kl(logq.double(), logp.double()).item()) < 1e-9
    assert metrics.jensen_shannon(1.0, 3.0) == 2.0
    # extreme weights: one sample dominates -> ESS = 1/N, no overflow
    big = logq.clone()
    big[0] -= 5000.0
    assert abs(metrics.compute_ess(logp, big) - 1.0 / n) < 1e-9
    assert math.isnan(metrics.compute_ess(logp[:0], logq[:0]))


def _worker(rank, world, port, n, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from normalizing_flows_amd import metrics, parallel
        logp, logq = _data(n)
        lo, hi = parallel.shard_bounds(n, rank, world)
        res = {'ess': metrics.compute_ess(logp[lo:hi], logq[lo:hi]),
               'fkl': metrics.forward_kl(logp[lo:hi], logq[lo:hi]),
               'rkl': metrics.reverse_kl(logq[lo:hi], logp[lo:hi])}
        if rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [2, 1001])
def test_two_rank_metrics_equal_single_process(tmp_path, n):
    from oracle import nf_oracle as O
    out = str(tmp_path / 'm.pt')
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n, out), nprocs=2, join=True)
    got = torch.load(out)
    logp, logq = _data(n)
    assert abs(got['ess'] - O.compute_ess(logp.double(), logq.double()).item()) < 1e-9
    assert abs(got['fkl'] - O.forward_kl(logp.double(), logq.double()).item()) < 1e-9
    assert abs(got['rkl'] - O.reverse_kl(logq.double(), logp.double()).item()) < 1e-9


def test_checkpoint_round_trip_and_plain_state_dict(tmp_path):
    import normalizing_flows_amd as nf
    from oracle import nf_oracle as O
    D, H, K = 6, 16, 4
    mk = lambda: nf.CompositeTransform([nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        D, H, num_bins=K, tails='linear', tail_bound=3.) for _ in range(2)])
    torch.manual_seed(1)
    a = mk()
    torch.manual_seed(2)
    b = mk()
    path = str(tmp_path / 'flow.pt')
    nf.save_checkpoint(a, path, step=12, note='toy')
    assert nf.load_checkpoint(b, path) == {'step': 12, 'note': 'toy'}
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    # a bare state_dict with the reference's key names (what the reference itself would save)
    sd = O.seeded_made_state_dict(D, H, 3 * K - 1, seed=3)
    plain = str(tmp_path / 'ref.pt')
    torch.save({'_transforms.0.' + k: v for k, v in sd.items()}, plain)
    assert nf.load_checkpoint(b, plain, strict=False) == {}
    assert torch.equal(b.state_dict()['_transforms.0.autoregressive_net.final_layer.weight'],
                       sd['autoregressive_net.final_layer.weight'])
    with pytest.raises(TypeError):
        nf.save_checkpoint(a, path, bad=torch.zeros(1))
