"""A few launches of ONE standalone elementwise kernel, for rocprofv3 (kernel trace or one --pmc pass):

    python3 benchmarks/run_elementwise.py rqs_fwd [--rows 65536] [--features 784] [--reps 3]

kernels: rqs_fwd rqs_inv rqs_bwd affine_fwd linear_fwd quadratic_fwd cubic_fwd cubic_bwd quadratic_bwd"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402,F401
from normalizing_flows_amd import _lib  # noqa: E402


def make(kernel, B, D, K=8, T=3.0):
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(B, D, device='cuda', generator=g)
    u = torch.rand(B, D, device='cuda', generator=g)
    y = torch.empty_like(x)
    ld = torch.empty(B, device='cuda')
    P = {'rqs': 3 * K - 1, 'affine': 2, 'linear': K, 'quadratic': 2 * K + 1, 'cubic': 2 * K + 2}[kernel.split('_')[0]]
    p = torch.randn(B, D * P, device='cuda', generator=g)
    gy, gl = torch.randn(B, D, device='cuda', generator=g), torch.randn(B, device='cuda', generator=g)
    fns = {
        'rqs_fwd': lambda: _lib.rqs(x, p, K, 'linear', T, out=y, logabsdet=ld),
        'rqs_inv': lambda: _lib.rqs(x, p, K, 'linear', T, inverse=True, out=y, logabsdet=ld),
        'rqs_bwd': lambda: _lib.rqs_bwd(x, p, gy, gl, K, 'linear', T),
        'affine_fwd': lambda: _lib.affine(x, p, _lib.NF_AFFINE_INTERLEAVED, out=y, logabsdet=ld),
        'linear_fwd': lambda: _lib.linear_spline(u, p, K, out=y, logabsdet=ld),
        'quadratic_fwd': lambda: _lib.quadratic_spline(u, p, K, out=y, logabsdet=ld),
        'cubic_fwd': lambda: _lib.cubic_spline(x, p, K, tails='linear', bound=10.0, out=y, logabsdet=ld),
        'quadratic_bwd': lambda: _lib.family_spline_bwd('quadratic', u, p, gy, gl, K),
        'cubic_bwd': lambda: _lib.family_spline_bwd('cubic', x, p, gy, gl, K, tails='linear', tail_bound=10.0),
    }
    return fns[kernel], P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('kernel')
    ap.add_argument('--rows', type=int, default=65536)
    ap.add_argument('--features', type=int, default=784)
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    fn, _ = make(a.kernel, a.rows, a.features)
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
