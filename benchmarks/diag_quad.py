import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import normalizing_flows_amd as nf
from conftest import load_golden
fx = load_golden('quadspline_K8_T3')
K, Tb = int(fx['num_bins']), float(fx['tail_bound'])
x = torch.from_numpy(fx['inv_x']).cuda().reshape(-1, 1)
y, ld, bins = nf._lib.quadratic_spline(x, torch.from_numpy(fx['params']).cuda(), K, tails='linear', tail_bound=Tb, inverse=True, want_bins=True)
for name, got, r64, r32 in (('y', y.reshape(-1), fx['inv_y64'], fx['inv_y32']), ('ld', ld, fx['inv_ld64'], fx['inv_ld32'])):
    e = (got.cpu().double() - torch.from_numpy(r64)).abs() / torch.from_numpy(r64).abs().clamp_min(1.0)
    i = int(e.argmax())
    print(name, 'worst', i, e[i].item(), 'got', got[i].item(), 'ref64', r64[i], 'ref32', r32[i], 'x', fx['inv_x'][i], 'bin', int(bins.reshape(-1)[i]), fx['inv_bins64'][i], fx['inv_bins32'][i])
