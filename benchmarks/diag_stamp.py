"""Phase timing of the fused final-layer kernel (library built with EXTRA=-DNF_STAMP only).

Each wave accumulates shader-clock cycles per K-loop phase and writes them where the bin indices
normally go: A = wait at the first barrier, S = wait for the prefetched operands + LDS stores,
B = second barrier, L = issue of the next prefetch, M = LDS reads + MFMAs, E = epilogue.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402

D, H, K, T = 784, 512, 8, 3.0
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
nf.set_matmul_dtype(sys.argv[2] if len(sys.argv) > 2 else 'f32')
torch.manual_seed(0)
t = nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
    features=D, hidden_features=H, num_bins=K, tails='linear', tail_bound=T, num_blocks=2).cuda().eval()
x = torch.randn(B, D, device='cuda')
with torch.no_grad():
    for _ in range(2):
        y, bins = t.forward_with_bins(x)[0], None
    out = t._transform_rows(x, x, None, torch.empty_like(x), None, False, want_bins=True)
torch.cuda.synchronize()
bins = out[2]
mode = sys.argv[2] if len(sys.argv) > 2 else 'f32'
m_tiles = (B + 127) // 128 if mode == 'bf16x3' else (B + 63) // 64
WAVES = 4
n_tiles = (D + 10) // 11
grid = m_tiles * n_tiles
s = bins.view(-1).view(torch.int64)[: grid * WAVES * 16].view(grid, WAVES, 16).cpu().double()
kt = s[:, :, 7]
names = ['barA', 'store', 'barB', 'load', 'mma', 'epi', '', '', 'e0bar', 'e1wr', 'e2eval', 'e3sum']
tot = s[:, :, :6].sum(2) + s[:, :, 8:12].sum(2)
print('blocks', grid, 'mean cycles per wave', tot.mean().item())
for i, n in enumerate(names):
    if not n:
        continue
    print('%6s total %6.1f%%   per k-tile %8.1f cycles' % (n, 100 * s[:, :, i].sum().item() / tot.sum().item(),
                                                          (s[:, :, i] / kt.clamp_min(1)).mean().item()))
full = kt[:, 0] == kt.max()
print('full-K blocks: %d' % full.sum().item())
for i, n in enumerate(names):
    if not n:
        continue
    print('%6s per k-tile (16-tile blocks) %8.1f' % (n, (s[full][:, :, i] / kt.max()).mean().item()))
start = s[:, :, 6]
print('kernel span cycles', (start.max() - start.min()).item())
for w in range(WAVES):
    print('wave', w, ' '.join('%s=%.0f' % (n, s[full][:, w, i].mean().item()) for i, n in enumerate(names) if n))
