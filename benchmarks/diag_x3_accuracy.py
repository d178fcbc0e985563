"""Error of each matmul mode against the fp64 fixtures (same checks the parity tests make, printed)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import normalizing_flows_amd as nf  # noqa: E402
from conftest import load_golden, golden_state_dict  # noqa: E402
from oracle import nf_oracle as O  # noqa: E402
import test_gpu_parity as tg  # noqa: E402


def q(got, ref):
    e = (got.double().cpu().reshape(-1) - torch.as_tensor(ref).double().reshape(-1)).abs() / torch.as_tensor(ref).double().reshape(-1).abs().clamp_min(1.0)
    return 'max %.2e q99 %.2e mean %.2e' % (e.max().item(), torch.quantile(e, 0.99).item(), e.mean().item())


for mode in ('f32', 'bf16x3'):
    nf.set_matmul_dtype(mode)
    fx = load_golden('rq_coupling_linear')
    comp = tg._coupling_stack(nf, fx, 'linear')
    with torch.no_grad():
        y, ld = comp(tg.dev(fx['x']))
    print(mode, 'coupling y ', q(y, fx['y64']), '| ref fp32', q(torch.as_tensor(fx['y']), fx['y64']))
    print(mode, 'coupling ld', q(ld, fx['ld64']), '| ref fp32', q(torch.as_tensor(fx['ld']), fx['ld64']))
    with torch.no_grad():
        xi, ldi = comp.inverse(tg.dev(fx['y']))
    print(mode, 'coupling inv x ', q(xi, fx['x']))
    print(mode, 'coupling inv ld', q(ldi, -torch.as_tensor(fx['ld64'])), '| ref fp32', q(torch.as_tensor(fx['inv_of_y_ld']), -torch.as_tensor(fx['ld64'])))
    fx = load_golden('rqar_cfg3')
    K, Tb, D, H = int(fx['num_bins']), float(fx['tail_bound']), int(fx['features']), int(fx['hidden'])
    sds = [O.seeded_made_state_dict(D, H, 3 * K - 1, seed=int(s)) for s in fx['seeds']]
    comp = nf.CompositeTransform(tg._load_layers(nf, sds, D, H, K, Tb)).cuda().eval()
    with torch.no_grad():
        y, ld = comp(tg.dev(fx['x']))
    print(mode, 'cfg3 y ', q(y, fx['y64']))
    print(mode, 'cfg3 ld', q(ld, fx['ld64']))
