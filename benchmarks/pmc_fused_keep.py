"""Runs the fused final-layer + spline forward with and without the kept-parameter output (for rocprofv3 --pmc passes:
L2 hit / miss counts of made_rqs_kernel<..., KEEP> against the plain launch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402
from normalizing_flows_amd import _lib  # noqa: E402

nf.set_matmul_dtype('bf16')
B, D, H, K = 32768, 784, 512, 8
layer = nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(features=D, hidden_features=H, num_bins=K,
                                                                   tails='linear', tail_bound=3.0).cuda()
net = layer.autoregressive_net
x = torch.randn(B, D, device='cuda')
with torch.no_grad():
    hidden = net.forward_hidden(x).contiguous()
    pf = net.final_layer._packed_fused_final(D, K, 'linear')
    for _ in range(3):
        _lib.made_rqs_params(hidden, pf, x, 3.0)
        _lib.made_rqs(hidden, pf, x, 3.0)
    torch.cuda.synchronize()
