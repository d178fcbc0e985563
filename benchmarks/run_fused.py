"""Runs one cfg3 layer's forward (2 hidden-stack launches + the fused final-layer kernel) a few times:
the target of rocprofv3 --pmc passes.   python3 benchmarks/run_fused.py [rows] [f32|bf16] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
nf.set_matmul_dtype(sys.argv[2] if len(sys.argv) > 2 else 'f32')
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(0)
t = nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
    features=784, hidden_features=512, num_bins=8, tails='linear', tail_bound=3.0, num_blocks=2).cuda().eval()
x = torch.randn(B, 784, device='cuda')
with torch.no_grad():
    for _ in range(reps):
        y, ld = t(x)
torch.cuda.synchronize()
print('ok', float(ld.mean()))
