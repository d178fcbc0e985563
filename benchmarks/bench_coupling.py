"""cfg-2 (BASELINE.json configs[1] shape): 8 x RQ-spline coupling layers, D=64, alternating masks,
ResidualNet(hidden 32, 2 blocks) conditioners, K=8, linear tails; log_prob samples/sec.

    python benchmarks/bench_coupling.py [--batch 1048576] [--tail-bound 3] [--steps 5] [--cpu-sample 131072]

Prints one JSON line with the GPU rate, the HBM roofline of the layer (3460 algorithmic bytes per
sample per layer) and, unless --no-cpu, the oracle timed on the host on a bounded sample + the parity error."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normalizing_flows_amd as nf  # noqa: E402

D, HID, K, L = 64, 32, 8, 8


def build(tail_bound):
    torch.manual_seed(0)
    layers = []
    for i in range(L):
        mask = torch.ones(D)
        mask[i % 2::2] = -1        # alternating binary mask (reference helper_functions.py:461-472)
        layers.append(nf.PiecewiseRationalQuadraticCouplingTransform(
            mask, lambda i_, o_: nf.ResidualNet(i_, o_, hidden_features=HID, num_blocks=2),
            num_bins=K, tails='linear', tail_bound=tail_bound))
    return nf.BaseNeuralSplineFlow(nf.CompositeTransform(layers), nf.StandardNormal([D]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=1048576)
    ap.add_argument('--tail-bound', type=float, default=3.0)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--cpu-sample', type=int, default=131072)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16x3', 'f16x2', 'bf16'])
    args = ap.parse_args()
    nf.set_matmul_dtype(args.dtype)
    flow = build(args.tail_bound)
    sd = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.cuda().eval()
    x = torch.randn(args.batch, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1234))
    with torch.no_grad():
        lp = flow.log_prob(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lp = flow.log_prob(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    bytes_per_sample = L * 3460
    out = {'metric': 'log_prob samples/sec, 8x RQ-coupling D=64 (cfg-2)', 'value': args.batch / dt, 'unit': 'samples/s',
           'ms_per_step': dt * 1e3, 'batch': args.batch, 'tail_bound': args.tail_bound, 'dtype': args.dtype,
           'roofline': {'bound': 'hbm', 'achieved': args.batch * bytes_per_sample / dt / 1e9, 'peak': 8000.0,
                        'unit': 'GB/s', 'frac': args.batch * bytes_per_sample / dt / 1e9 / 8000.0,
                        'algorithmic_bytes_per_sample': bytes_per_sample}}
    if not args.no_cpu:
        from oracle import nf_oracle as O
        n = min(args.cpu_sample, args.batch)
        xc = x[:n].cpu()
        tsd = {k[len('_transform.'):]: v for k, v in sd.items() if k.startswith('_transform.')}
        funcs = []
        for j in range(L):
            s = O.split_state_dict(tsd, j)
            funcs.append(lambda v, s=s: O.rq_coupling(s, v, K, 'linear', args.tail_bound, hidden_features=HID))
        best = float('inf')
        with torch.no_grad():
            for _ in range(3):
                t0 = time.perf_counter()
                ref = O.flow_log_prob(funcs, xc)
                best = min(best, time.perf_counter() - t0)
        err = ((lp[:n].cpu().double() - ref.double()).abs() / ref.double().abs().clamp_min(1.0)).max().item()
        out['cpu_baseline'] = {'value': n / best, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                               'sample': 'oracle on the first %d samples, best of 3' % n}
        out['parity_max_rel_err_vs_oracle'] = err
    print(json.dumps(out))


if __name__ == '__main__':
    main()
