"""Weight-gradient GEMM of one masked layer: fp32-MFMA kernel vs the split-operand (f16x2) one.

    python benchmarks/bench_wgrad.py [--rows 32768] [--inp 1024] [--out 18032] [--reps 5] [--only split|f32]

The mask is MADE's for a final layer over `inp` hidden units (degrees 1 .. D-1 cyclic) and out = D * 23 parameters."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402,F401
from normalizing_flows_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=32768)
    ap.add_argument('--inp', type=int, default=1024)
    ap.add_argument('--out', type=int, default=18032)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    B, I, O = a.rows, a.inp, a.out
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(B, I, device='cuda', generator=g)
    gy = torch.randn(B, O, device='cuda', generator=g) * 1e-4
    D = max(O // 23, 2)
    deg_in = torch.arange(I, device='cuda') % (D - 1) + 1
    deg_out = torch.arange(O, device='cuda') // max(O // D, 1) + 1
    mask = (deg_out[:, None] > deg_in[None, :]).float()
    ktab = _lib.PackedLinear(torch.zeros(O, I, device='cuda'), mask, None, _lib.NF_DTYPE_F32).ktab
    xm, gm = _lib.maxabs(x), _lib.maxabs(gy)
    res = {'rows': B, 'in': I, 'out': O, 'unmasked_frac': round(mask.mean().item(), 3)}
    flop = 2.0 * B * I * O * mask.mean().item()

    def timed(fn):
        import time
        t0 = time.time()
        while time.time() - t0 < 0.5:        # clocks ramp up over the first few hundred ms of a fresh process
            fn()
            torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / a.reps

    if a.only in ('', 'split'):
        ms = timed(lambda: _lib.masked_linear_split_bwd_weight(x, gy, mask, ktab, True, x_max=xm, grad_y_max=gm))
        res['split_ms'] = round(ms, 3)
        res['split_unmasked_tflops'] = round(flop / ms / 1e9, 1)
    if a.only in ('', 'half'):
        ms = timed(lambda: _lib.masked_linear_split_bwd_weight(x, gy, mask, ktab, True, x_max=xm, grad_y_max=gm, terms=1))
        res['half_ms'] = round(ms, 3)
        res['half_unmasked_tflops'] = round(flop / ms / 1e9, 1)
    if a.only in ('', 'f32'):
        ms = timed(lambda: _lib.masked_linear_bwd_weight(x, gy, mask, ktab, True))
        res['f32_ms'] = round(ms, 3)
        res['f32_unmasked_tflops'] = round(flop / ms / 1e9, 1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
