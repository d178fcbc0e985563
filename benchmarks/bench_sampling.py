"""Inverse-sampling throughput of the cfg-3 RQ-spline MAF (secondary to bench.py's log_prob metric).

    python benchmarks/bench_sampling.py [--batch 65536] [--layers 5] [--dtype f32|bf16]

Times flow.forward(z) = transform.inverse(z) (reference distributions.py:192-193): per layer the
incremental inverse runs 513 passes (one hidden tile per layer + one fused final tile each) instead of
the reference's 784 full passes; checks the round trip through log_prob's direction afterwards."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=65536)
    ap.add_argument('--layers', type=int, default=5)
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--windowed', action='store_true', help='the windowed incremental inverse instead of the row-resident kernel')
    args = ap.parse_args()
    nf.set_matmul_dtype(args.dtype)
    if args.windowed:
        from normalizing_flows_amd import transforms as tr
        tr.ROW_RESIDENT_INVERSE = False
    torch.manual_seed(0)
    D, H, K = 784, 512, 8
    layers = [nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(D, H, num_bins=K, tails='linear', tail_bound=3.)
              for _ in range(args.layers)]
    flow = nf.BaseNeuralSplineFlow(nf.CompositeTransform(layers), nf.StandardNormal([D])).cuda().eval()
    z = torch.randn(args.batch, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(7))
    with torch.no_grad():
        x, ld = flow.forward(z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            x, ld = flow.forward(z)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        z2, ld2 = flow.reverse(x)
    print(json.dumps({'metric': 'inverse sampling samples/sec, RQ-spline MAF D=784', 'value': args.batch / dt,
                      'unit': 'samples/s', 'seconds': dt, 'batch': args.batch, 'layers': args.layers,
                      'dtype': args.dtype, 'path': 'windowed' if args.windowed else 'row-resident', 'passes_per_layer': len(layers[0]._feature_groups()),
                      'roundtrip_max_abs_err': (z2 - z).abs().max().item(),
                      'logdet_cancel_max_abs': (ld + ld2).abs().max().item()}))


if __name__ == '__main__':
    main()
