"""Phase timing of the split-operand fused kernel (library built with -DNF_STAMP, benchmarks/build_variant.sh stamp
-DNF_STAMP; run with NF_HIP_LIB=.../variants/libnf_hip_stamp.so).  Wave 0..3 of every workgroup write 8 clock stamps
where the bin indices normally go: start, inputs staged, K loop done, (tile dropped, splines done) x 2 passes, end."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402

D, H, K, T = 784, 512, 8, 3.0
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
nf.set_matmul_dtype(sys.argv[2] if len(sys.argv) > 2 else 'f16x2')
torch.manual_seed(0)
t = nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
    features=D, hidden_features=H, num_bins=K, tails='linear', tail_bound=T, num_blocks=2).cuda().eval()
x = torch.randn(B, D, device='cuda')
with torch.no_grad():
    for _ in range(2):
        t(x)
    out = t._transform_rows(x, x, None, torch.empty_like(x), None, False, want_bins=True)
torch.cuda.synchronize()
bins = out[2]
WAVES = int(os.environ.get('NF_DIAG_WAVES', '8'))
grid = ((B + 127) // 128) * ((D + 10) // 11)
s = bins.view(-1).view(torch.int64)[: grid * WAVES * 16].view(grid, WAVES, 16).cpu()
st = s[:, :, :8].double()
steps = s[:, 0, 8].double()
names = ['inputs', 'kloop', 'drop0', 'eval0', 'drop1', 'eval1', 'tail']
d = st[:, :, 1:] - st[:, :, :-1]
tot = st[:, :, 7] - st[:, :, 0]
print('workgroups', grid, 'mean clock ticks per workgroup (wave mean)', tot.mean().item(), 'K steps mean', steps.mean().item())
for i, n in enumerate(names):
    print('%8s %9.1f ticks  %5.1f%%' % (n, d[:, :, i].mean().item(), 100 * d[:, :, i].sum().item() / tot.sum().item()))
if WAVES == 8:
    print('K loop per step: waiting for own DMA pieces %.1f ticks, at the barrier %.1f, rest (issue, LDS reads, MFMA) %.1f' % (
        (s[:, :, 11].double().sum() / (steps.sum() * WAVES)).item(), (s[:, :, 12].double().sum() / (steps.sum() * WAVES)).item(),
        ((d[:, :, 1].sum() - s[:, :, 11].double().sum() - s[:, :, 12].double().sum()) / (steps.sum() * WAVES)).item()))
full = steps == steps.max()
print('full-K workgroups: %d, ticks per K step there %.1f' % (full.sum().item(), (d[full][:, :, 1].mean() / steps.max()).item()))
for i, n in enumerate(names):
    print('%8s (full-K) %9.1f' % (n, d[full][:, :, i].mean().item()))
span = (st[:, :, 7].max() - st[:, :, 0].min()).item()
print('launch span in ticks', span, '-> ticks per microsecond if the launch took 5.63 ms: %.1f' % (span / 5630.0))
hw = s[:, 0, 9]
print('wave slot (HW_ID & 15) histogram of wave 0:', torch.bincount((hw & 15).long(), minlength=16).tolist())
print('SIMD id histogram:', torch.bincount(((hw >> 4) & 3).long(), minlength=4).tolist())
first = torch.arange(grid) < 512
for slot in (0, 1):
    sel = first & ((hw & 15) == slot)
    print('first-round workgroups in wave slot %d: %d, inputs phase %.0f ticks, whole %.0f' %
          (slot, sel.sum().item(), d[sel][:, :, 0].mean().item(), tot[sel].mean().item()))
# co-residency: for every workgroup, the fraction of its K loop during which its CU partner (same CU, other slot)
# was also inside a K loop.  CU identity = HW_ID bits (cu 8..11, sh 12, se 13..15, xcc from a separate read is absent:
# blocks b and b+8 share an XCD, so group by (block % 8, se, sh, cu)
cu_key = ((hw >> 8) & 0xff) * 8 + (torch.arange(grid) % 8)
k0, k1 = st[:, 0, 1], st[:, 0, 2]
import collections
by_cu = collections.defaultdict(list)
for i in range(grid):
    by_cu[int(cu_key[i])].append(i)
print('distinct CU keys', len(by_cu))
import bisect
rows = []
for key, idx in by_cu.items():
    a = [(k0[i].item(), k1[i].item(), int(hw[i] & 15), steps[i].item()) for i in idx]
    for slot in (0, 1):
        mine = sorted(t for t in a if t[2] == slot)
        other = sorted(t for t in a if t[2] != slot)
        ob = [t[0] for t in other]
        for (b, e, _, n) in mine:
            if n < 8:
                continue
            ov = 0.0
            j = max(0, bisect.bisect_left(ob, b) - 2)
            while j < len(other) and other[j][0] < e:
                ov += max(0.0, min(e, other[j][1]) - max(b, other[j][0]))
                j += 1
            rows.append((ov / (e - b), (e - b) / n))
rows = torch.tensor(rows)
print('K-loop ticks per step by the fraction of the loop the CU partner spent in ITS K loop:')
for lo in (0.0, 0.1, 0.3, 0.5, 0.7, 0.9):
    hi = {0.0: 0.1, 0.1: 0.3, 0.3: 0.5, 0.5: 0.7, 0.7: 0.9, 0.9: 1.01}[lo]
    sel = (rows[:, 0] >= lo) & (rows[:, 0] < hi)
    if sel.any():
        print('  overlap %.1f-%.1f: %6d workgroups, %7.1f ticks per step' % (lo, hi, sel.sum().item(), rows[sel, 1].mean().item()))
