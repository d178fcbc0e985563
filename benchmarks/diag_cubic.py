import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import normalizing_flows_amd as nf
from conftest import load_golden
for tag in ('K8_T3', 'K6_pm10', 'K16_T5'):
    fx = load_golden('cubspline_' + tag)
    K, Tb, tails = int(fx['num_bins']), float(fx['tail_bound']), 'linear' if int(fx['tails']) else None
    x = torch.from_numpy(fx['inv_x']).cuda().reshape(-1, 1)
    y, ld, bins = nf._lib.cubic_spline(x, torch.from_numpy(fx['params']).cuda(), K, tails=tails, bound=Tb, inverse=True, want_bins=True)
    bad = torch.isnan(y.reshape(-1)) | torch.isnan(ld)
    idx = bad.nonzero().reshape(-1).cpu().tolist()
    print(tag, 'nan at', idx[:10], 'ref y64', [float(fx['inv_y64'][i]) for i in idx[:10]], 'ref ld64', [float(fx['inv_ld64'][i]) for i in idx[:10]], 'x', [float(fx['inv_x'][i]) for i in idx[:10]], 'y', [float(y.reshape(-1)[i]) for i in idx[:10]])
