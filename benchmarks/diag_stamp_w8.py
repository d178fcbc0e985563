"""Phase timing of the f16x2 eight / sixteen-wave fused kernel (library built with -DNF_STAMP:
benchmarks/build_fused_variant.sh stamp -DNF_STAMP; run with NF_HIP_LIB=.../variants/libnf_hip_stamp.so and
NF_FUSED_F16X2_KERNEL=w8|w16|t8).  Every wave writes its clock stamps where the bin indices normally go: start, prologue issued,
K loop done, epilogue passes done, end; plus the K steps of its tile and the ticks it waited for its own DMA pieces
and at the step barrier.      python benchmarks/diag_stamp_w8.py [rows = 131072]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402

D, H, K, T = 784, 512, 8, 3.0
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
flavour = os.environ.get('NF_FUSED_F16X2_KERNEL', 'w8')
rt = 1 if flavour in ('w8', 't4') else 2
nf.set_matmul_dtype('f16x2')
torch.manual_seed(0)
t = nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
    features=D, hidden_features=H, num_bins=K, tails='linear', tail_bound=T, num_blocks=2).cuda().eval()
x = torch.randn(B, D, device='cuda')
with torch.no_grad():
    for _ in range(30):                      # (settle the clock)
        t(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = t._transform_rows(x, x, None, torch.empty_like(x), None, False, want_bins=True)
torch.cuda.synchronize()
bins = out[2]
NW = {'w8': 8, 'w16': 16, 't8': 8, 't4': 4}[flavour]
grid = ((B + 128 * rt - 1) // (128 * rt)) * ((D + 10) // 11)
s = bins.view(-1).view(torch.int64)[: grid * NW * 16].view(grid, NW, 16).cpu()
st = s[:, :, :5].double()
steps = s[:, 0, 8].double()
names = ['prologue', 'kloop', 'epilogue', 'tail']
d = st[:, :, 1:] - st[:, :, :-1]
tot = st[:, :, 4] - st[:, :, 0]
print(flavour, 'rows per workgroup', 128 * rt, 'workgroups', grid, 'ticks per workgroup (wave mean) %.0f' % tot.mean().item(),
      'K steps mean %.2f' % steps.mean().item())
for i, n in enumerate(names):
    print('%9s %9.1f ticks  %5.1f%%' % (n, d[:, :, i].mean().item(), 100 * d[:, :, i].sum().item() / tot.sum().item()))
nsteps = steps.sum() * NW
print('K loop per step: waiting for own DMA pieces %.1f ticks, at the barrier %.1f, rest (issue, LDS reads, MFMA) %.1f; total %.1f' % (
    (s[:, :, 11].double().sum() / nsteps).item(), (s[:, :, 12].double().sum() / nsteps).item(),
    ((d[:, :, 1].sum() - s[:, :, 11].double().sum() - s[:, :, 12].double().sum()) / nsteps).item(),
    (d[:, :, 1].sum() / nsteps).item()))
full = steps == steps.max()
print('full-K workgroups: %d, ticks per K step there %.1f, epilogue there %.0f' % (
    full.sum().item(), (d[full][:, :, 1].mean() / steps.max()).item(), d[full][:, :, 2].mean().item()))
span = (st[:, :, 4].max() - st[:, :, 0].min()).item()
print('launch span in ticks %.0f; sum over workgroups of their life / (256 CUs x span) = %.2f resident workgroups per CU'
      % (span, tot.mean().item() * grid / (256 * span)))
rt_ticks = s[:, :, 13].double()          # s_memrealtime: 100 MHz
ok = rt_ticks > 0
print('in-kernel clock: %.3f GHz (median over waves of shader ticks / 100 MHz ticks)' % (
    (tot[ok] / rt_ticks[ok]).median().item() * 0.1))
print('per CU: %.2f M shader cycles of workgroup life per 256 CUs and resident workgroup -> launch time at that clock: %.2f ms'
      % (tot.mean().item() * grid / 256 / (2 if rt == 1 else 1) / 1e6,
         tot.mean().item() * grid / 256 / (2 if rt == 1 else 1) / ((tot[ok] / rt_ticks[ok]).median().item() * 1e8) * 1e3))
