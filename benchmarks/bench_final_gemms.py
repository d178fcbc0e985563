"""Forward and input-gradient GEMM of the MADE final layer (hidden -> D * P) on the split-operand kernel.

    python benchmarks/bench_final_gemms.py [--rows 32768] [--hidden 512] [--features 784] [--dtype bf16|f16x2|bf16x3]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402,F401
from normalizing_flows_amd import _lib, made  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=32768)
    ap.add_argument('--hidden', type=int, default=512)
    ap.add_argument('--features', type=int, default=784)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    B, H, D, P = a.rows, a.hidden, a.features, 23
    O = D * P
    dt = made._MATMUL_DTYPE[a.dtype]
    g = torch.Generator(device='cuda').manual_seed(1)
    w = torch.randn(O, H, device='cuda', generator=g) * 0.05
    deg_in = torch.arange(H, device='cuda') % (D - 1) + 1
    deg_out = torch.arange(O, device='cuda') // P + 1
    mask = (deg_out[:, None] > deg_in[None, :]).float()
    h = torch.randn(B, H, device='cuda', generator=g)
    gy = torch.randn(B, O, device='cuda', generator=g) * 1e-3
    fwd = _lib.PackedLinearSplit(w, mask, torch.zeros(O, device='cuda'), dt)
    bwd = _lib.PackedLinearSplit(w, mask, None, dt, transposed=True)
    hm, gm = _lib.maxabs(h), _lib.maxabs(gy)
    y = torch.empty(B, O, device='cuda')
    gx = torch.empty(B, H, device='cuda')

    def timed(fn):
        t0 = time.time()
        while time.time() - t0 < 0.5:
            fn()
            torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / a.reps

    live = mask.mean().item()
    res = {'rows': B, 'hidden': H, 'out': O, 'dtype': a.dtype, 'unmasked_frac': round(live, 3)}
    ms = timed(lambda: _lib.masked_linear_split(h, fwd, relu_in=True, out=y, x_max=hm))
    res['fwd_ms'] = round(ms, 3)
    res['fwd_unmasked_tflops'] = round(2.0 * B * H * O * live / ms / 1e9, 1)
    res['fwd_write_gbs'] = round(B * O * 4 / ms / 1e6, 1)
    ms = timed(lambda: _lib.masked_linear_split(gy, bwd, out=gx, x_max=gm, gate=h, reverse_k=True))
    res['bwd_input_ms'] = round(ms, 3)
    res['bwd_input_unmasked_tflops'] = round(2.0 * B * H * O * live / ms / 1e9, 1)
    res['bwd_input_read_gbs'] = round(B * O * 4 / ms / 1e6, 1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
