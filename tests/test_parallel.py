"""The N>1 path on CPU: two gloo ranks shard a batch, evaluate their shard (here with the CPU oracle
standing in for the per-rank kernel work, which needs no communication) and all-reduce the
[sum log p, count] pair; the result must equal the single-process mean."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, n, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from normalizing_flows_amd import parallel
        from oracle import nf_oracle as O
        torch.set_num_threads(2)
        D, H, K = 12, 24, 4
        sd = O.seeded_made_state_dict(D, H, 3 * K - 1, seed=7)
        x = torch.randn(n, D, generator=torch.Generator().manual_seed(3))

        def log_prob(v):
            return O.flow_log_prob([lambda u: O.rq_autoregressive_forward(sd, u, K, 3.0)], v)

        lp, lo, hi, mean = parallel.sharded_log_prob(log_prob, x)
        assert lp.shape[0] == hi - lo
        if rank == 0:
            torch.save({'mean': mean, 'lo': lo, 'hi': hi}, out)
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_exactly():
    from normalizing_flows_amd import parallel
    for n in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('n', [64, 37])
def test_two_rank_log_likelihood_reduction(tmp_path, n):
    from oracle import nf_oracle as O
    out = str(tmp_path / 'r0.pt')
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n, out), nprocs=2, join=True)
    got = torch.load(out)
    D, H, K = 12, 24, 4
    sd = O.seeded_made_state_dict(D, H, 3 * K - 1, seed=7)
    x = torch.randn(n, D, generator=torch.Generator().manual_seed(3))
    want = O.flow_log_prob([lambda u: O.rq_autoregressive_forward(sd, u, K, 3.0)], x).double().mean().item()
    assert abs(got['mean'] - want) < 1e-5 * max(1.0, abs(want))
    assert (got['lo'], got['hi']) == (0, (n + 1) // 2)


def _grad_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from normalizing_flows_amd import parallel
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 300, 7, 1024)]
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        nb = parallel.allreduce_gradients(params, bucket_bytes=1024)
        if rank == 0:
            torch.save({'grads': [p.grad.clone() for p in params], 'buckets': nb}, out)
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    out = str(tmp_path / 'g0.pt')
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_grad_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got['buckets'] >= 2                      # several buckets in flight
    for i, g in enumerate(got['grads']):
        assert torch.allclose(g, torch.full_like(g, 1.5 * (i + 1)))      # mean of ranks' (1, 2) x (i+1)


def _stream_worker(rank, world, port, n, rows_per_pass, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from normalizing_flows_amd import parallel
        from oracle import nf_oracle as O
        torch.set_num_threads(2)
        D, H, K = 12, 24, 4
        sd = O.seeded_made_state_dict(D, H, 3 * K - 1, seed=7)
        x = torch.randn(n, D, generator=torch.Generator().manual_seed(3))
        passes = []

        def make_rows(s, e):
            passes.append((s, e))
            return x[s:e]

        def log_prob(v):
            return O.flow_log_prob([lambda u: O.rq_autoregressive_forward(sd, u, K, 3.0)], v)

        total, count, span = parallel.streamed_log_likelihood(log_prob, n, make_rows, rows_per_pass)
        gen = torch.Generator().manual_seed(100 + rank)

        def sample(m):
            z = torch.randn(m, 3, generator=gen) * 2.0 + 1.0
            return z, -z.pow(2).sum(1)

        st = parallel.streamed_sampling(sample, n, rows_per_pass)
        torch.save({'total': total, 'count': count, 'span': span, 'passes': passes, 'st_count': st['count'],
                    'st_mean': st['mean'], 'st_var': st['var'], 'st_logq': st['mean_logq']}, out % rank)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n,rows_per_pass', [(101, 16), (5, 64), (1, 8)])
def test_two_rank_streamed_log_likelihood(tmp_path, n, rows_per_pass):
    """cfg-4's loop (parallel.streamed_log_likelihood) on two gloo ranks: shards streamed in short passes with a
    ragged last pass (and, for n = 1, an EMPTY shard on rank 1) reduce to the single-process sum."""
    from oracle import nf_oracle as O
    out = str(tmp_path / 's%d.pt')
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_stream_worker, args=(2, port, n, rows_per_pass, out), nprocs=2, join=True)
    got = [torch.load(out % r) for r in range(2)]
    D, H, K = 12, 24, 4
    sd = O.seeded_made_state_dict(D, H, 3 * K - 1, seed=7)
    x = torch.randn(n, D, generator=torch.Generator().manual_seed(3))
    want = O.flow_log_prob([lambda u: O.rq_autoregressive_forward(sd, u, K, 3.0)], x).double().sum().item()
    for r in range(2):
        assert got[r]['count'] == n and abs(got[r]['total'] - want) < 1e-6 * max(1.0, abs(want))
        lo, hi = got[r]['span']
        assert got[r]['passes'] == [(s, min(hi, s + rows_per_pass)) for s in range(lo, hi, rows_per_pass)]
        assert got[r]['st_count'] == n
    assert got[0]['span'][1] == got[1]['span'][0] and got[1]['span'][1] == n
    assert torch.equal(got[0]['st_mean'], got[1]['st_mean']) and got[0]['st_mean'].shape == (3,)
    if n > 50:
        assert (got[0]['st_mean'] - 1.0).abs().max() < 1.0 and (got[0]['st_var'] - 4.0).abs().max() < 2.5


def _dp_model(use_bn):
    torch.manual_seed(11)
    layers = [torch.nn.Linear(6, 16)]
    if use_bn:
        layers.append(torch.nn.BatchNorm1d(16, eps=1e-3))
    layers += [torch.nn.ReLU(), torch.nn.Linear(16, 16)]
    if use_bn:
        layers.append(torch.nn.BatchNorm1d(16, eps=1e-3))
    layers += [torch.nn.ReLU(), torch.nn.Linear(16, 4)]
    return torch.nn.Sequential(*layers).double()


def _dp_loss(model, x):
    return (model(x) - x[:, :4].sin()).pow(2).sum(1).mean()


def _dp_worker(rank, world, port, use_bn, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from normalizing_flows_amd import parallel
        torch.set_num_threads(1)
        model = parallel.convert_sync_batchnorm(_dp_model(use_bn)).train()
        x = torch.randn(64, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
        lo, hi = parallel.shard_bounds(64, rank, world)
        sync = parallel.GradientSynchronizer(model.parameters(), bucket_bytes=1024)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        early = []
        for _ in range(3):
            sync.zero_grad()
            _dp_loss(model, x[lo:hi]).backward()
            early.append(sync.finish())
            opt.step()
        if rank == 0:
            torch.save({'sd': model.state_dict(), 'early': early, 'buckets': len(sync.buckets)}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('use_bn', [False, True])
def test_two_rank_data_parallel_steps_equal_the_single_process_steps(tmp_path, use_bn):
    """Three Adam steps on 2 x 32 samples (gloo ranks; gradients through GradientSynchronizer's hook-driven bucket
    all-reduces, BatchNorm through SyncBatchNorm1d) leave the weights, and the BatchNorm running statistics, where
    three single-process steps on the 64 samples leave them (fp64)."""
    out = str(tmp_path / 'dp.pt')
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(2, port, use_bn, out), nprocs=2, join=True)
    got = torch.load(out)
    model = _dp_model(use_bn).train()
    x = torch.randn(64, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        _dp_loss(model, x).backward()
        opt.step()
    want = model.state_dict()
    assert list(got['sd'].keys()) == list(want.keys())             # SyncBatchNorm1d keeps BatchNorm1d's keys
    for k, v in want.items():
        # a bias in front of a BatchNorm has an exactly zero gradient in exact arithmetic; what backward leaves there is
        # rounding noise (1e-17), which Adam's g / (sqrt(v) + 1e-8) turns into steps of ~1e-11 whose direction differs
        # between the two summation orders: those entries are compared at 1e-9 absolute instead
        pre_bn_bias = use_bn and k in ('0.bias', '3.bias', '1.running_mean', '4.running_mean')    # (the means carry those biases)
        assert torch.allclose(got['sd'][k].double(), v.double(), rtol=1e-9, atol=1e-9 if pre_bn_bias else 1e-11), \
            (k, (got['sd'][k].double() - v.double()).abs().max().item())
    assert got['buckets'] >= 2 and all(e >= 1 for e in got['early'])   # collectives started before backward returned


def test_gradient_synchronizer_single_process_views():
    """Without a process group the synchronizer only provides flat gradient views: backward accumulates in place,
    zero_grad keeps the views, and a replaced .grad is folded back into its bucket."""
    from normalizing_flows_amd import parallel
    model = _dp_model(False)
    sync = parallel.GradientSynchronizer(model.parameters(), bucket_bytes=512)
    x = torch.randn(8, 6, dtype=torch.float64)
    sync.zero_grad()
    _dp_loss(model, x).backward()
    assert sync.finish() == 0
    ref = [p.grad.clone() for p in model.parameters()]
    flat = torch.cat([b['flat'] for b in sync.buckets])
    assert flat.abs().sum() > 0 and all(p.grad.data_ptr() == sync._bucket_of[p][1].data_ptr() for p in model.parameters())
    model.zero_grad(set_to_none=True)                              # a caller that drops the views
    _dp_loss(model, x).backward()
    sync.finish()
    for p, r in zip(model.parameters(), ref):
        assert torch.equal(p.grad, r) and p.grad.data_ptr() == sync._bucket_of[p][1].data_ptr()


def _bn_worker(rank, world, port, affine, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from normalizing_flows_amd import parallel
        torch.set_num_threads(1)
        x, g = _bn_data()
        lo, hi = parallel.shard_bounds(x.shape[0], rank, world)
        bn = parallel.convert_sync_batchnorm(torch.nn.Sequential(_bn_module(affine)))[0].train()
        xs = x[lo:hi].clone().requires_grad_(True)
        y = bn(xs)
        (y * g[lo:hi]).sum().backward()
        torch.save({'y': y.detach(), 'gx': xs.grad, 'mean': bn.running_mean, 'var': bn.running_var,
                    'gw': None if not affine else bn.weight.grad, 'gb': None if not affine else bn.bias.grad}, out % rank)
    finally:
        dist.destroy_process_group()


def _bn_data():
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(50, 5, generator=gen)
    x[:, 0] = x[:, 0] * 0.1 + 100.0           # |mean| >> std: E[x^2] - mean^2 in fp32 keeps no digit of this variance
    x[:, 1] = x[:, 1] * 1e-2 - 3000.0
    return x, torch.randn(50, 5, generator=gen)


def _bn_module(affine):
    torch.manual_seed(4)
    bn = torch.nn.BatchNorm1d(5, eps=1e-3, affine=affine)
    if affine:
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-1, 1)
    return bn


@pytest.mark.parametrize('affine', [True, False])
def test_two_rank_sync_batchnorm_with_a_large_mean_feature(tmp_path, affine):
    """fp32 SyncBatchNorm1d on 2 ranks x 25 rows against nn.BatchNorm1d on the 50 rows, with features whose mean is
    1e3-3e5 standard deviations from zero (the uncentred fp32 variance collapsed to 0 there), and with
    affine=False (weight / bias None)."""
    out = str(tmp_path / 'bn%d.pt')
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_bn_worker, args=(2, port, affine, out), nprocs=2, join=True)
    got = [torch.load(out % r) for r in range(2)]
    x, g = _bn_data()
    bn = _bn_module(affine).double().train()             # the float64 truth; fp32 inputs are exact in it
    xr = x.double().requires_grad_(True)
    y = bn(xr)
    (y * g.double()).sum().backward()
    y_got = torch.cat([got[0]['y'], got[1]['y']]).double()
    gx_got = torch.cat([got[0]['gx'], got[1]['gx']]).double()
    # what fp32 can hold: x - mean is exact, the mean itself is rounded to fp32 (half an ulp of |mean|), and that
    # offset is divided by sqrt(var + eps); everything else is 1e-6 relative
    sd = (x.double().var(0, unbiased=False) + 1e-3).sqrt()
    tol = 1e-5 + 2.0 ** -23 * x.double().mean(0).abs() / sd * (1.5 if affine else 1.0)
    assert ((y_got - y.detach()).abs().max(0).values <= tol).all(), ((y_got - y.detach()).abs().max(0).values, tol)
    assert torch.allclose(gx_got, xr.grad, rtol=1e-3, atol=1e-3 * xr.grad.abs().max().item())
    for r in range(2):
        assert torch.allclose(got[r]['mean'].double(), bn.running_mean, rtol=1e-6, atol=1e-6)
        assert torch.allclose(got[r]['var'].double(), bn.running_var, rtol=1e-5, atol=1e-8)
    assert torch.equal(got[0]['mean'], got[1]['mean']) and torch.equal(got[0]['var'], got[1]['var'])
    if affine:                                 # parameter gradients stay local sums; their sum is the global gradient
        assert torch.allclose((got[0]['gw'] + got[1]['gw']).double(), bn.weight.grad, rtol=1e-3, atol=2e-2)
        assert torch.allclose((got[0]['gb'] + got[1]['gb']).double(), bn.bias.grad, rtol=1e-4, atol=1e-4)


def _accum_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from normalizing_flows_amd import parallel
        torch.set_num_threads(1)
        model = _dp_model(False)
        x = torch.randn(64, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
        lo, hi = parallel.shard_bounds(64, rank, world)
        mid = (lo + hi) // 2
        sync = parallel.GradientSynchronizer(model.parameters(), bucket_bytes=1024)
        sync.zero_grad()
        _dp_loss(model, x[lo:mid]).backward()
        raised = False
        try:
            _dp_loss(model, x[mid:hi]).backward()          # a second backward with the collectives under way
        except RuntimeError as e:
            raised = 'no_sync' in str(e)
        sync.finish()
        sync.zero_grad()
        with sync.no_sync():
            _dp_loss(model, x[lo:mid]).backward()
        _dp_loss(model, x[mid:hi]).backward()
        sync.finish()
        if rank == 0:
            torch.save({'raised': raised, 'grads': [p.grad.clone() for p in model.parameters()]}, out)
    finally:
        dist.destroy_process_group()


def test_gradient_synchronizer_second_backward_raises_and_no_sync_accumulates(tmp_path):
    out = str(tmp_path / 'acc.pt')
    port = 39500 + (os.getpid() % 2000)
    mp.spawn(_accum_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got['raised']
    model = _dp_model(False)
    x = torch.randn(64, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    # four micro-batches of 16 rows: each rank sums its two, the all-reduce averages the ranks
    total = sum(_dp_loss(model, x[s:s + 16]) for s in range(0, 64, 16)) / 2
    total.backward()
    for g, p in zip(got['grads'], model.parameters()):
        assert torch.allclose(g, p.grad, rtol=1e-10, atol=1e-12)
