"""HBM roofline of the standalone elementwise kernels: RQ spline (forward / inverse), affine (autoregressive
interleaved / coupling blocked parameters) and the linear / quadratic / cubic spline families.

    python benchmarks/bench_elementwise.py [--rows 65536] [--features 784] [--reps 10]

achieved = algorithmic bytes per launch (parameters read + x read + y written + logdet written)
/ the launch's average duration from HIP events on the launch stream; peak = 8 TB/s."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402
from normalizing_flows_amd import _lib  # noqa: E402

PEAK_GBS = 8000.0


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=65536)
    ap.add_argument('--features', type=int, default=784)
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    B, D, K, T = args.rows, args.features, 8, 3.0
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(B, D, device='cuda', generator=g)
    u = torch.rand(B, D, device='cuda', generator=g)
    y = torch.empty_like(x)
    ld = torch.empty(B, device='cuda')
    out = {'rows': B, 'features': D, 'peak_gbs': PEAK_GBS, 'kernels': {}}

    def report(name, P, fn):
        ms = timed(fn, args.reps)
        byts = B * (D * (4 * P + 8) + 4)
        gbs = byts / (ms * 1e-3) / 1e9
        out['kernels'][name] = {'params_per_element': P, 'algorithmic_bytes': byts, 'ms': round(ms, 4),
                                'achieved_gbs': round(gbs, 1), 'frac': round(gbs / PEAK_GBS, 3)}

    p = torch.randn(B, D * (3 * K - 1), device='cuda', generator=g)
    report('rqs_fwd_K8_linear_tails', 3 * K - 1, lambda: _lib.rqs(x, p, K, 'linear', T, out=y, logabsdet=ld))
    report('rqs_inv_K8_linear_tails', 3 * K - 1, lambda: _lib.rqs(x, p, K, 'linear', T, inverse=True, out=y, logabsdet=ld))
    del p
    p = torch.randn(B, D * 2, device='cuda', generator=g)
    report('affine_fwd_interleaved', 2, lambda: _lib.affine(x, p, _lib.NF_AFFINE_INTERLEAVED, out=y, logabsdet=ld))
    report('affine_inv_interleaved', 2, lambda: _lib.affine(x, p, _lib.NF_AFFINE_INTERLEAVED, inverse=True, out=y,
                                                            logabsdet=ld))
    report('affine_fwd_blocked', 2, lambda: _lib.affine(x, p, _lib.NF_AFFINE_BLOCKED, out=y, logabsdet=ld))
    del p
    p = torch.randn(B, D * K, device='cuda', generator=g)
    report('linear_spline_fwd_K8', K, lambda: _lib.linear_spline(u, p, K, out=y, logabsdet=ld))
    del p
    p = torch.randn(B, D * (2 * K + 1), device='cuda', generator=g)
    report('quadratic_spline_fwd_K8', 2 * K + 1, lambda: _lib.quadratic_spline(u, p, K, out=y, logabsdet=ld))
    del p
    p = torch.randn(B, D * (2 * K + 2), device='cuda', generator=g)
    report('cubic_spline_fwd_K8', 2 * K + 2, lambda: _lib.cubic_spline(x, p, K, tails='linear', bound=10.0, out=y, logabsdet=ld))
    del p

    # backward kernels: parameters read + their gradients written + x, grad_y read + grad_x written + grad_logabsdet
    def report_bwd(name, P, fn):
        ms = timed(fn, args.reps)
        byts = B * (D * (8 * P + 12) + 4)
        gbs = byts / (ms * 1e-3) / 1e9
        out['kernels'][name] = {'params_per_element': P, 'algorithmic_bytes': byts, 'ms': round(ms, 4),
                                'achieved_gbs': round(gbs, 1), 'frac': round(gbs / PEAK_GBS, 3)}

    gy = torch.randn(B, D, device='cuda', generator=g)
    gl = torch.randn(B, device='cuda', generator=g)
    p = torch.randn(B, D * (3 * K - 1), device='cuda', generator=g)
    report_bwd('rqs_bwd_K8_linear_tails', 3 * K - 1, lambda: _lib.rqs_bwd(x, p, gy, gl, K, 'linear', T))
    del p
    p = torch.randn(B, D * (2 * K + 2), device='cuda', generator=g)
    report_bwd('cubic_spline_bwd_K8', 2 * K + 2, lambda: _lib.family_spline_bwd('cubic', x, p, gy, gl, K, tails='linear',
                                                                                tail_bound=10.0))
    del p
    p = torch.randn(B, D * K, device='cuda', generator=g)
    report_bwd('linear_spline_bwd_K8', K, lambda: _lib.family_spline_bwd('linear', u, p, gy, gl, K))
    del p
    p = torch.randn(B, D * (2 * K + 1), device='cuda', generator=g)
    report_bwd('quadratic_spline_bwd_K8', 2 * K + 1, lambda: _lib.family_spline_bwd('quadratic', u, p, gy, gl, K))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
