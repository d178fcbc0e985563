"""Golden fixtures for the linear and quadratic spline families, generated from the REFERENCE
(models/splines/linear.py, models/splines/quadratic.py).  Same rules as gen_golden.py: runs only in the
build container, writes small .npz files that are committed.

    python tests/golden/gen_golden_splines.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.modules["wandb"] = types.ModuleType("wandb")
import matplotlib  # noqa: E402
matplotlib.use("Agg")
sys.path.insert(0, "/root/reference")
import helper_functions  # noqa: E402  (must precede models.*)
from models.splines import linear as ref_lin  # noqa: E402
from models.splines import quadratic as ref_quad  # noqa: E402
from models.splines import cubic as ref_cub  # noqa: E402

_last = {}
_orig = helper_functions.searchsorted


def _rec(bin_locations, inputs, eps=1e-6):
    idx = _orig(bin_locations, inputs, eps)
    _last['idx'] = idx.clone()
    return idx


helper_functions.searchsorted = _rec


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('wrote', name, {k: v.shape for k, v in out.items() if v.ndim})


def inputs_for(N, T, g, tails):
    if tails:
        x = torch.randn(N, generator=g) * (0.6 * T)
        edge = torch.tensor([-T, T, 0.0, 2 * T, -2 * T, np.nextafter(np.float32(T), np.float32(0)),
                             np.nextafter(np.float32(T), np.float32(1e9))], dtype=torch.float32)
    else:
        x = torch.rand(N, generator=g) * 0.998 + 0.001
        edge = torch.tensor([0.0, 1.0, 0.5], dtype=torch.float32)
    x[:len(edge)] = edge
    return x, len(edge)


def gen_linear():
    for (K, T, tails, tag) in [(8, 3.0, True, 'K8_T3'), (5, 1.0, False, 'K5_unit'), (32, 50.0, True, 'K32_T50')]:
        g = torch.Generator().manual_seed(300 + K)
        N = 4096
        u = torch.randn(N, K, generator=g)
        u[N // 2:] *= 3.0
        arrays = {'params': u, 'num_bins': K, 'tail_bound': T, 'tails': int(tails)}
        for inverse, key in [(False, 'fwd'), (True, 'inv')]:
            x, ne = inputs_for(N, T, g, tails)
            outs = {}
            for dt, sfx in ((torch.float32, '32'), (torch.float64, '64')):
                _last.clear()
                if tails:
                    y, ld = ref_lin.unconstrained_linear_spline(x.to(dt), u.to(dt).clone(), inverse=inverse, tail_bound=T)
                else:
                    y, ld = ref_lin.linear_spline(x.to(dt), u.to(dt).clone(), inverse=inverse)
                outs['y' + sfx], outs['ld' + sfx] = y, ld
            arrays.update({key + '_x': x, key + '_n_edge': ne, **{key + '_' + k: v for k, v in outs.items()}})
        save('linspline_' + tag, **arrays)


def gen_quadratic():
    for (K, T, tails, tag) in [(8, 3.0, True, 'K8_T3'), (6, 1.0, False, 'K6_unit'), (16, 10.0, True, 'K16_T10')]:
        g = torch.Generator().manual_seed(400 + K)
        N = 4096
        nh = K - 1 if tails else K + 1
        params = torch.randn(N, K + nh, generator=g)
        params[N // 2:] *= 2.0
        uw, uh = params[:, :K], params[:, K:]
        arrays = {'params': params, 'num_bins': K, 'tail_bound': T, 'tails': int(tails)}
        for inverse, key in [(False, 'fwd'), (True, 'inv')]:
            x, ne = inputs_for(N, T, g, tails)
            outs = {}
            for dt, sfx in ((torch.float32, '32'), (torch.float64, '64')):
                _last.clear()
                if tails:
                    y, ld = ref_quad.unconstrained_quadratic_spline(x.to(dt), uw.to(dt).clone(), uh.to(dt).clone(),
                                                                    inverse=inverse, tail_bound=T)
                    inside = (x >= -T) & (x <= T)
                    bins = torch.full((N,), -1, dtype=torch.int64)
                    bins[inside] = _last['idx']
                else:
                    y, ld = ref_quad.quadratic_spline(x.to(dt), uw.to(dt).clone(), uh.to(dt).clone(), inverse=inverse)
                    bins = _last['idx']
                outs['y' + sfx], outs['ld' + sfx], outs['bins' + sfx] = y, ld, bins.to(torch.int16)
            arrays.update({key + '_x': x, key + '_n_edge': ne, **{key + '_' + k: v for k, v in outs.items()}})
        save('quadspline_' + tag, **arrays)


def gen_cubic():
    for (K, T, tails, tag) in [(8, 3.0, True, 'K8_T3'), (6, 10.0, False, 'K6_pm10'), (16, 5.0, True, 'K16_T5')]:
        g = torch.Generator().manual_seed(500 + K)
        N = 4096
        params = torch.randn(N, 2 * K + 2, generator=g)
        params[N // 2:] *= 2.0
        uw, uh, ul, ur = params[:, :K], params[:, K:2 * K], params[:, 2 * K:2 * K + 1], params[:, 2 * K + 1:]
        arrays = {'params': params, 'num_bins': K, 'tail_bound': T, 'tails': int(tails)}
        for inverse, key in [(False, 'fwd'), (True, 'inv')]:
            x = torch.randn(N, generator=g) * (0.6 * T)
            edge = torch.tensor([-T, T, 0.0] + ([2 * T, -2 * T] if tails else []), dtype=torch.float32)
            if not tails:
                x = x.clamp(-T, T)
            x[:len(edge)] = edge
            outs = {}
            for dt, sfx in ((torch.float32, '32'), (torch.float64, '64')):
                _last.clear()
                args = [v.to(dt).clone() for v in (uw, uh, ul, ur)]
                if tails:
                    y, ld = ref_cub.unconstrained_cubic_spline(x.to(dt), *args, inverse=inverse, tail_bound=T)
                    inside = (x >= -T) & (x <= T)
                    bins = torch.full((N,), -1, dtype=torch.int64)
                    bins[inside] = _last['idx']
                else:
                    y, ld = ref_cub.cubic_spline(x.to(dt), *args, inverse=inverse)
                    bins = _last['idx']
                outs['y' + sfx], outs['ld' + sfx], outs['bins' + sfx] = y, ld, bins.to(torch.int16)
            arrays.update({key + '_x': x, key + '_n_edge': len(edge), **{key + '_' + k: v for k, v in outs.items()}})
        save('cubspline_' + tag, **arrays)


if __name__ == '__main__':
    which = sys.argv[1:] or ['linear', 'quadratic', 'cubic']
    if 'linear' in which:
        gen_linear()
    if 'quadratic' in which:
        gen_quadratic()
    if 'cubic' in which:
        gen_cubic()
