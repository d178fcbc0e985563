"""Generates the golden fixtures in this directory from the REFERENCE implementation.

Runs only in the build container (needs /root/reference); the fixtures it writes are
committed and are what pins oracle/nf_oracle.py and the HIP path.  Nothing here is
loaded at test time.

    python tests/golden/gen_golden.py

The reference is imported with an in-memory stub for the absent `wandb` module (never
called on this path); `helper_functions` must be imported before `models.*`
(SURVEY.md 8c).  Big-model weights are NOT stored: they come from
oracle.nf_oracle.seeded_made_state_dict(seed) (a CPU torch.Generator, identical on
every machine) and are loaded into the reference modules with load_state_dict(strict).
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
from models import made, transforms, flows  # noqa: E402
from models.nn import resnet  # noqa: E402
from models.splines import rational_quadratic as ref_rq  # noqa: E402

from oracle import nf_oracle as O  # noqa: E402

torch.set_num_threads(8)

# ---- capture the bin indices the reference computes (wrap, do not modify) -------------
_last_bins = {}
_orig_search = helper_functions.searchsorted


def _recording_search(bin_locations, inputs, eps=1e-6):
    idx = _orig_search(bin_locations, inputs, eps)
    _last_bins['idx'] = idx.clone()
    return idx


helper_functions.searchsorted = _recording_search


class f64:
    """Run the reference in a float64 default (its CPU policy, main.py:21-24); needed because
    CompositeTransform allocates its accumulator in the default dtype (transforms.py:24)."""
    def __enter__(self):
        torch.set_default_dtype(torch.float64)

    def __exit__(self, *a):
        torch.set_default_dtype(torch.float32)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-40s %8.1f KB' % (name, os.path.getsize(path) / 1024))


def sd_np(sd, prefix=''):
    return {prefix + k.replace('.', '/'): v for k, v in sd.items()}


def ref_unconstrained(x, uw, uh, ud, inverse, T, mins=(1e-3, 1e-3, 1e-3)):
    inside = (x >= -T) & (x <= T)
    y, ld = ref_rq.unconstrained_rational_quadratic_spline(
        x, uw.clone(), uh.clone(), ud.clone(), inverse=inverse, tails='linear', tail_bound=T,
        min_bin_width=mins[0], min_bin_height=mins[1], min_derivative=mins[2])
    bins = torch.full(x.shape, -1, dtype=torch.int64)
    bins[inside] = _last_bins['idx']
    return y, ld, bins


def far_from_knots(x, u, lo, hi, min_size, tol):
    """True where x is farther than tol from every knot built (in fp64) from u."""
    cum, _ = O._knots(u.double(), min_size, lo, hi)
    return ((x.double()[..., None] - cum).abs() > tol).all(dim=-1)


# ======================================================================= F1: spline alone
def gen_spline():
    for (K, T, N, tag) in [(4, 1.0, 4096, 'K4_T1'), (8, 3.0, 4096, 'K8_T3'), (32, 50.0, 1024, 'K32_T50')]:
        g = torch.Generator().manual_seed(100 + K)
        P = 3 * K - 1
        params = torch.randn(N, P, generator=g)
        params[N // 2:] *= 3.0
        uw, uh, ud = params[:, :K], params[:, K:2 * K], params[:, 2 * K:]
        arrays = {'params': params, 'num_bins': K, 'tail_bound': T}
        for inverse, key in [(False, 'fwd'), (True, 'inv')]:
            x = torch.randn(N, generator=g) * (0.6 * T)
            # forced edge cases: the bounds themselves, just inside/outside, zero
            edge = torch.tensor([-T, np.nextafter(np.float32(-T), np.float32(0)),
                                 np.nextafter(np.float32(-T), np.float32(-1e9)),
                                 np.nextafter(np.float32(T), np.float32(0)),
                                 np.nextafter(np.float32(T), np.float32(1e9)),
                                 0.0, 2 * T, -2 * T], dtype=torch.float32)
            if np.float32(T) + np.float32(1e-6) > np.float32(T):
                edge = torch.cat([edge, torch.tensor([T], dtype=torch.float32)])
            x[:len(edge)] = edge
            # resample anything within tol of an interior knot so bin_idx is well defined
            u = uh if inverse else uw
            tol = 2e-5 * T
            for _ in range(20):
                ok = far_from_knots(x, u, -T, T, 1e-3, tol)
                ok[:len(edge)] = True
                if ok.all():
                    break
                x[~ok] = torch.randn(int((~ok).sum()), generator=g) * (0.6 * T)
            y32, ld32, bins = ref_unconstrained(x, uw, uh, ud, inverse, T)
            y64, ld64, bins64 = ref_unconstrained(x.double(), uw.double(), uh.double(), ud.double(), inverse, T)
            body = slice(len(edge), None)
            assert (bins[body] == bins64[body]).all(), 'fp32/fp64 reference bins disagree away from knots'
            arrays.update({key + '_x': x, key + '_y32': y32, key + '_ld32': ld32,
                           key + '_y64': y64, key + '_ld64': ld64,
                           key + '_bins': bins.to(torch.int16), key + '_n_edge': len(edge)})
        save('spline_' + tag, **arrays)

    # tails=None on [0,1] (the coupling path may call this, transforms.py:227-229): K+1 derivatives
    K, N = 8, 2048
    g = torch.Generator().manual_seed(77)
    params = torch.randn(N, 3 * K + 1, generator=g)
    uw, uh, ud = params[:, :K], params[:, K:2 * K], params[:, 2 * K:]
    arrays = {'params': params, 'num_bins': K}
    for inverse, key in [(False, 'fwd'), (True, 'inv')]:
        x = torch.rand(N, generator=g) * 0.998 + 0.001
        u = uh if inverse else uw
        for _ in range(20):
            ok = far_from_knots(x, u, 0., 1., 1e-3, 2e-5)
            if ok.all():
                break
            x[~ok] = torch.rand(int((~ok).sum()), generator=g) * 0.998 + 0.001
        y32, ld32 = ref_rq.rational_quadratic_spline(x, uw.clone(), uh.clone(), ud.clone(), inverse=inverse)
        bins = _last_bins['idx'].clone()
        y64, ld64 = ref_rq.rational_quadratic_spline(x.double(), uw.double(), uh.double(), ud.double(), inverse=inverse)
        assert (bins == _last_bins['idx']).all()
        arrays.update({key + '_x': x, key + '_y32': y32, key + '_ld32': ld32, key + '_y64': y64,
                       key + '_ld64': ld64, key + '_bins': bins.to(torch.int16)})
    save('spline_K8_notails', **arrays)


# ======================================================================= F2: MADE
def gen_made():
    # (a) small, the reference's own default init, everything stored incl. masks/degrees
    for tag, kw in [('res', dict()), ('ff', dict(use_residual_blocks=False)),
                    ('bn', dict(use_batch_norm=True)), ('ffbn', dict(use_residual_blocks=False, use_batch_norm=True)),
                    ('nb3', dict(num_blocks=3))]:
        torch.manual_seed(11)
        net = made.MADE(features=6, hidden_features=16, output_multiplier=23, **kw)
        if kw.get('use_batch_norm'):
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.normal_(0, 0.5)
                    m.running_var.uniform_(0.5, 2.0)
                    m.weight.data.uniform_(0.5, 1.5)
                    m.bias.data.normal_(0, 0.2)
        net.eval()
        x = torch.randn(32, 6)
        with torch.no_grad():
            y32 = net(x)
            y64 = net.double()(x.double())
        net.float()
        save('made_small_' + tag, x=x, y32=y32, y64=y64, **sd_np(net.state_dict(), 'sd/'))

    # context-conditioned MADE
    torch.manual_seed(12)
    net = made.MADE(features=6, hidden_features=16, context_features=5, output_multiplier=2).eval()
    x, c = torch.randn(32, 6), torch.randn(32, 5)
    with torch.no_grad():
        y32 = net(x, c)
    save('made_small_ctx', x=x, context=c, y32=y32, **sd_np(net.state_dict(), 'sd/'))

    # (b) larger shapes: seeded weights (not stored), masks pinned by row sums
    for (D, H, mult, B, scale, tag) in [(64, 128, 23, 64, None, 'D64_H128_m23'),
                                        (64, 128, 23, 64, 0.1, 'D64_H128_m23_n01'),
                                        (784, 512, 23, 8, None, 'D784_H512_m23'),
                                        (784, 512, 2, 16, None, 'D784_H512_m2'),
                                        (5, 32, 23, 16, None, 'D5_H32_m23'),
                                        (1, 8, 23, 16, None, 'D1_H8_m23')]:
        seed = 1000 + D + H + mult
        sd = O.seeded_made_state_dict(D, H, mult, seed=seed, scale=scale, prefix='')
        net = made.MADE(features=D, hidden_features=H, output_multiplier=mult).eval()
        for k, v in net.state_dict().items():          # the oracle's masks must BE the reference's
            if k.endswith('.mask') or k.endswith('.degrees'):
                assert torch.equal(v.to(sd[k].dtype), sd[k]), k
        net.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, D, generator=g)
        with torch.no_grad():
            y32 = net(x)
            y64 = net.double()(x.double())
        rows = {('masksum/' + k.replace('.', '/')): v.sum(dim=1).to(torch.int32)
                for k, v in sd.items() if k.endswith('.mask')}
        save('made_' + tag, x=x, y32=y32, y64=y64, seed=seed, features=D, hidden=H, mult=mult,
             scale=-1.0 if scale is None else scale, **rows)


def gen_random_mask():
    """`random_mask=True` (made.py:134-141: hidden degrees drawn with torch.randint; feed-forward blocks only,
    :305-306): the MADE alone and an RQ-spline transform on it, forward and the D-pass inverse.  The drawn masks and
    degrees travel in the state_dict; `build_seed` is the torch seed the networks were constructed under, so a port
    that draws its degrees with the same calls can be checked to build the same masks."""
    build_seed = 77
    torch.manual_seed(build_seed)
    net = made.MADE(features=6, hidden_features=16, output_multiplier=23, use_residual_blocks=False,
                    random_mask=True, num_blocks=2).eval()
    x = torch.randn(32, 6)
    with torch.no_grad():
        y32 = net(x)
        y64 = net.double()(x.double())
    net.float()
    save('made_small_random_mask', x=x, y32=y32, y64=y64, build_seed=build_seed, **sd_np(net.state_dict(), 'sd/'))

    D, H, K, T = 6, 16, 8, 3.0
    torch.manual_seed(build_seed + 1)
    t = transforms.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=D, hidden_features=H, num_bins=K, tails='linear', tail_bound=T, num_blocks=2,
        use_residual_blocks=False, random_mask=True).eval()
    with torch.no_grad():
        for prm in t.parameters():                       # default init is near-identity: make the spline do something
            prm.mul_(3.0)
    x = torch.randn(48, D) * 1.5
    with torch.no_grad():
        y32, l32 = t(x)
        sd32 = {k: v.clone() for k, v in t.state_dict().items()}
        with f64():
            t.double()
            y64, l64 = t(x.double())
            xi64, li64 = t.inverse(y64)
        t.float()
    save('rqar_random_mask', x=x, y32=y32, lad32=l32, y64=y64, lad64=l64, xinv64=xi64, ladinv64=li64,
         build_seed=build_seed + 1, features=D, hidden=H, num_bins=K, tail_bound=T, **sd_np(sd32, 'sd/'))


def gen_activations():
    """MADE / ResidualNet built with activations other than the default ReLU (made.py:302, resnet.py:66)."""
    for name, act in [('tanh', torch.tanh), ('elu', torch.nn.functional.elu)]:
        for tag, kw in [('res', dict()), ('ffbn', dict(use_residual_blocks=False, use_batch_norm=True))]:
            torch.manual_seed(13)
            net = made.MADE(features=6, hidden_features=16, output_multiplier=23, activation=act, **kw)
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.normal_(0, 0.5)
                    m.running_var.uniform_(0.5, 2.0)
            with torch.no_grad():
                for p_ in net.parameters():
                    p_.mul_(1.5)
            net.eval()
            x = torch.randn(32, 6)
            with torch.no_grad():
                y32 = net(x)
                y64 = net.double()(x.double())
            net.float()
            save('made_small_%s_%s' % (tag, name), x=x, y32=y32, y64=y64, **sd_np(net.state_dict(), 'sd/'))
        torch.manual_seed(14)
        rn = resnet.ResidualNet(5, 12, hidden_features=32, num_blocks=2, activation=act).eval()
        with torch.no_grad():
            for p_ in rn.parameters():
                p_.mul_(1.5)
            x = torch.randn(32, 5)
            y32 = rn(x)
            y64 = rn.double()(x.double())
        rn.float()
        save('resnet_small_%s' % name, x=x, y32=y32, y64=y64, **sd_np(rn.state_dict(), 'sd/'))


# ======================================================================= F3: RQ-AR layers
def rq_ar_ref(D, H, K, T, sd):
    t = transforms.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=D, hidden_features=H, num_bins=K, tails='linear', tail_bound=T, num_blocks=2).eval()
    t.load_state_dict(sd, strict=True)
    return t


def gen_rq_ar():
    # small: reference default init, stored whole; fwd, bins, inverse (D passes)
    torch.manual_seed(21)
    D, H, K, T, L = 16, 32, 8, 3.0, 3
    layers = [transforms.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=D, hidden_features=H, num_bins=K, tails='linear', tail_bound=T) for _ in range(L)]
    comp = transforms.CompositeTransform(layers).eval()
    with torch.no_grad():
        for p in comp.parameters():           # push away from identity
            p.mul_(2.0)
    x = torch.randn(48, D) * 1.3
    with torch.no_grad():
        y, ld = comp(x)
        y1, ld1 = layers[0](x)
        params = layers[0].autoregressive_net(x).view(48, D, -1)
        _, _, bins1 = ref_unconstrained(x, params[..., :K], params[..., K:2 * K], params[..., 2 * K:], False, T)
        xi, ldi = comp.inverse(y)
        z = torch.randn(48, D)
        xs, lds = comp.inverse(z)
        c64 = comp.double()
        with f64():
            y64, ld64 = c64(x.double())
            xs64, lds64 = c64.inverse(z.double())
        comp.float()
    logp = torch.distributions.Normal(0., 1.).log_prob(y).sum(1) + ld
    save('rqar_small', x=x, y=y, ld=ld, y1=y1, ld1=ld1, bins1=bins1.to(torch.int16), inv_of_y=xi, inv_of_y_ld=ldi,
         z=z, xs=xs, lds=lds, y64=y64, ld64=ld64, xs64=xs64, lds64=lds64, logp=logp,
         num_bins=K, tail_bound=T, features=D, hidden=H, **sd_np(comp.state_dict(), 'sd/'))

    # cfg-3 shape: 5 layers D=784 H=512 K=8 T=3, seeded weights
    D, H, K, T, L, B = 784, 512, 8, 3.0, 5, 32
    seeds = [3000 + j for j in range(L)]
    refs = [rq_ar_ref(D, H, K, T, O.seeded_made_state_dict(D, H, 3 * K - 1, seed=s)) for s in seeds]
    comp = transforms.CompositeTransform(refs).eval()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, D, generator=g)
    with torch.no_grad():
        y, ld = comp(x)
        y1, ld1 = refs[0](x)
        params = refs[0].autoregressive_net(x).view(B, D, -1)
        _, _, bins1 = ref_unconstrained(x, params[..., :K], params[..., K:2 * K], params[..., 2 * K:], False, T)
        zi = torch.randn(4, D, generator=g)
        xs1, lds1 = refs[0].inverse(zi)               # one layer, 784 passes
        c64 = comp.double()
        with f64():
            y64, ld64 = c64(x.double())
            xs1_64, lds1_64 = refs[0].inverse(zi.double())
        comp.float()
    logp = torch.distributions.Normal(0., 1.).log_prob(y).sum(1) + ld
    logp64 = torch.distributions.Normal(0., 1.).log_prob(y64).sum(1) + ld64
    save('rqar_cfg3', x=x, y=y, ld=ld, y1=y1, ld1=ld1, bins1=bins1.to(torch.int8), zi=zi, xs1=xs1, lds1=lds1,
         y64=y64, ld64=ld64, xs1_64=xs1_64, lds1_64=lds1_64, logp=logp, logp64=logp64,
         seeds=np.array(seeds), num_bins=K, tail_bound=T, features=D, hidden=H)


# ======================================================================= F4: affine AR + coupling
def gen_affine_coupling():
    torch.manual_seed(41)
    D, H = 64, 128
    t = transforms.MaskedAffineAutoregressiveTransform(features=D, hidden_features=H).eval()
    x = torch.randn(64, D)
    z = torch.randn(8, D)
    with torch.no_grad():
        y, ld = t(x)
        xs, lds = t.inverse(z)
        y64, ld64 = t.double()(x.double())
        t.float()
    save('affine_ar', x=x, y=y, ld=ld, z=z, xs=xs, lds=lds, y64=y64, ld64=ld64, features=D, hidden=H,
         **sd_np(t.state_dict(), 'sd/'))

    D, K, T, Hn, L = 64, 8, 3.0, 32, 4
    for tails, tag in [('linear', 'linear'), (None, 'notails')]:
        torch.manual_seed(42)
        layers = []
        for j in range(L):
            mask = helper_functions.create_alternating_binary_mask(D, even=(j % 2 == 0))
            layers.append(transforms.PiecewiseRationalQuadraticCouplingTransform(
                mask=mask,
                transform_net_create_fn=lambda i, o: resnet.ResidualNet(i, o, hidden_features=Hn, num_blocks=2),
                num_bins=K, tails=tails, tail_bound=T))
        comp = transforms.CompositeTransform(layers).eval()
        with torch.no_grad():
            for p in comp.parameters():
                p.mul_(1.5)
        if tails is None:
            # one layer only: outputs of a [0,1] spline stay in [0,1] but keep the test simple
            comp = transforms.CompositeTransform(layers[:1]).eval()
            x = torch.rand(256, D) * 0.98 + 0.01
        else:
            x = torch.randn(256, D) * 1.5
        with torch.no_grad():
            y, ld = comp(x)
            xi, ldi = comp.inverse(y)
            with f64():
                y64, ld64 = comp.double()(x.double())
            comp.float()
        save('rq_coupling_' + tag, x=x, y=y, ld=ld, inv_of_y=xi, inv_of_y_ld=ldi, y64=y64, ld64=ld64,
             num_bins=K, tail_bound=T, hidden=Hn, features=D, n_layers=len(comp._transforms),
             **sd_np(comp.state_dict(), 'sd/'))

    torch.manual_seed(43)
    layers = []
    for j in range(3):
        mask = helper_functions.create_alternating_binary_mask(D, even=(j % 2 == 0))
        layers.append(transforms.AffineCouplingTransform(
            mask=mask, transform_net_create_fn=lambda i, o: resnet.ResidualNet(i, o, hidden_features=Hn, num_blocks=2)))
    comp = transforms.CompositeTransform(layers).eval()
    with torch.no_grad():
        for p in comp.parameters():
            p.mul_(1.5)
    x = torch.randn(256, D)
    with torch.no_grad():
        y, ld = comp(x)
        xi, ldi = comp.inverse(y)
        with f64():
            y64, ld64 = comp.double()(x.double())
        comp.float()
    save('affine_coupling', x=x, y=y, ld=ld, inv_of_y=xi, inv_of_y_ld=ldi, y64=y64, ld64=ld64, hidden=Hn,
         features=D, n_layers=3, **sd_np(comp.state_dict(), 'sd/'))


def gen_unconditional_coupling():
    """RQ coupling layers with apply_unconditional_transform=True (transforms.py:196-206: a PiecewiseRationalQuadraticCDF,
    :249-318, on the identity half), odd feature count so the halves differ; forward and inverse."""
    D, K, T, Hn, L = 9, 6, 3.0, 32, 3
    for tails, tag in [('linear', 'linear'), (None, 'notails')]:
        torch.manual_seed(45)
        layers = []
        for j in range(L):
            mask = helper_functions.create_alternating_binary_mask(D, even=(j % 2 == 0))
            layers.append(transforms.PiecewiseRationalQuadraticCouplingTransform(
                mask=mask,
                transform_net_create_fn=lambda i, o: resnet.ResidualNet(i, o, hidden_features=Hn, num_blocks=2),
                num_bins=K, tails=tails, tail_bound=T, apply_unconditional_transform=True))
        if tails is None:
            comp = transforms.CompositeTransform(layers[:1]).eval()
            x = torch.rand(128, D) * 0.98 + 0.01
        else:
            comp = transforms.CompositeTransform(layers).eval()
            x = torch.randn(128, D) * 1.5
        with torch.no_grad():
            for n_, p in comp.named_parameters():
                if 'unconditional' not in n_:
                    p.mul_(1.5)
            y, ld = comp(x)
            xi, ldi = comp.inverse(y)
            with f64():
                y64, ld64 = comp.double()(x.double())
                xi64, ldi64 = comp.inverse(y64)
            comp.float()
        save('rq_coupling_uncond_' + tag, x=x, y=y, ld=ld, inv_of_y=xi, inv_of_y_ld=ldi, y64=y64, ld64=ld64,
             inv_of_y64=xi64, inv_of_y64_ld=ldi64, num_bins=K, tail_bound=T, hidden=Hn, features=D,
             n_layers=len(comp._transforms), **sd_np(comp.state_dict(), 'sd/'))


def gen_cubic_unconditional():
    """Cubic coupling layers with apply_unconditional_transform=True (transforms.py:367-375: a PiecewiseCubicCDF,
    :417-473, on the identity half), linear tails (with tails=None the reference's CDF calls the dangling name
    `splines.cubic_spline`); forward and inverse."""
    D, K, T, Hn, L = 9, 6, 3.0, 32, 2
    torch.manual_seed(46)
    layers = []
    for j in range(L):
        mask = helper_functions.create_alternating_binary_mask(D, even=(j % 2 == 0))
        layers.append(transforms.PiecewiseCubicCouplingTransform(
            mask=mask,
            transform_net_create_fn=lambda i, o: resnet.ResidualNet(i, o, hidden_features=Hn, num_blocks=2),
            num_bins=K, tails='linear', tail_bound=T, apply_unconditional_transform=True))
    comp = transforms.CompositeTransform(layers).eval()
    x = torch.randn(128, D) * 1.5
    with torch.no_grad():
        for n_, p in comp.named_parameters():
            if 'unconditional' not in n_:
                p.mul_(1.5)
        y, ld = comp(x)
        xi, ldi = comp.inverse(y)
        with f64():
            y64, ld64 = comp.double()(x.double())
            xi64, ldi64 = comp.inverse(y64)
        comp.float()
    save('cubic_coupling_uncond_linear', x=x, y=y, ld=ld, inv_of_y=xi, inv_of_y_ld=ldi, y64=y64, ld64=ld64,
         inv_of_y64=xi64, inv_of_y64_ld=ldi64, num_bins=K, tail_bound=T, hidden=Hn, features=D,
         n_layers=L, **sd_np(comp.state_dict(), 'sd/'))
    try:
        bad = transforms.PiecewiseCubicCDF(shape=[3], num_bins=K)
        bad(torch.rand(4, 3))
        print('PiecewiseCubicCDF(tails=None) ran in the reference')
    except Exception as e:      # noqa: BLE001
        print('PiecewiseCubicCDF(tails=None) in the reference raises %s: %s' % (type(e).__name__, e))


# ======================================================================= F5: legacy MAF (fp64)
def gen_maf():
    torch.set_default_dtype(torch.float64)            # what main.py:21-24 selects on CPU
    try:
        for D in (2, 5):
            torch.manual_seed(50 + D)
            layers = [flows.MaskedAutoregressiveFlow(D, [64], reverse_input=(j % 2 == 0)) for j in range(5)]
            arrays = {'dim': D, 'n_layers': 5}
            z = torch.randn(128, D)
            h = z
            logJ = torch.zeros(128)
            with torch.no_grad():
                for j, f in enumerate(layers):
                    arrays['l%d/reverse_input' % j] = int(f.reverse_input)
                    lin = [m for m in f.made.model if isinstance(m, made.SimpleMaskedLinear)]
                    for i, m in enumerate(lin):
                        arrays['l%d/w%d' % (j, i)] = m.weight
                        arrays['l%d/b%d' % (j, i)] = m.bias
                        arrays['l%d/m%d' % (j, i)] = m.mask
                    h, lj = f.forward(h)
                    arrays['l%d/fwd_x' % j] = h
                    arrays['l%d/fwd_logJ' % j] = lj
                    logJ = logJ + lj
                arrays.update(z=z, x=h, logJ=logJ)
                r = h
                rl = torch.zeros(128)
                for j in reversed(range(5)):
                    r, lj = layers[j].reverse(r)
                    arrays['l%d/rev_z' % j] = r
                    arrays['l%d/rev_logJ' % j] = lj
                    rl = rl + lj
                arrays.update(rev_z=r, rev_logJ=rl)
            save('maf_D%d' % D, **arrays)
    finally:
        torch.set_default_dtype(torch.float32)


# ======================================================================= cfg-1: short training of the legacy MAF
def gen_maf_training():
    """BASELINE.json configs[0]: 5 MaskedAutoregressiveFlow(2, [64]) layers, float64, trained on two-moons data by
    the reference's own loop - Adam(lr 5e-3) + CosineAnnealingLR(250) for 250 steps of 2**14 fresh samples
    (test2.py:24-27, 127-135; model.py:233-244), loss = -mean of the density model.py:103-108 assigns to the batch
    (`apply_reverse_to_samples` with the SimpleNormal prior: forward KL up to the data entropy).  The data definition
    is this build's (normalizing-flows_amd/data.py, a CPU generator: the same points on every machine)."""
    from models import distributions as ref_dist
    import importlib.util
    spec = importlib.util.spec_from_file_location('nf_data', os.path.join(ROOT, 'normalizing-flows_amd', 'data.py'))
    nf_data = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nf_data)
    torch.set_default_dtype(torch.float64)
    try:
        D, L, steps, B, lr = 2, 5, 250, 2 ** 14, 5e-3
        torch.manual_seed(71)
        layers = torch.nn.ModuleList([flows.MaskedAutoregressiveFlow(D, [64], reverse_input=(j % 2 == 0))
                                      for j in range(L)])
        prior = ref_dist.SimpleNormal(torch.zeros(D), torch.ones(D))
        arrays = {'dim': D, 'n_layers': L, 'steps': steps, 'batch': B, 'lr': lr, 'data_seed': 2024}
        for j, f in enumerate(layers):
            arrays['l%d/reverse_input' % j] = int(f.reverse_input)
            lin = [m for m in f.made.model if isinstance(m, made.SimpleMaskedLinear)]
            for i, m in enumerate(lin):
                arrays['l%d/w%d' % (j, i)] = m.weight.detach().clone()
                arrays['l%d/b%d' % (j, i)] = m.bias.detach().clone()
                arrays['l%d/m%d' % (j, i)] = m.mask

        def log_density(x):                            # model.py:103-108, logq = 0
            logq = 0
            for layer in reversed(layers):
                x, logJ = layer.reverse(x)
                logq = logq + logJ.squeeze()
            return prior.log_prob(x) + logq

        x_eval = nf_data.two_moons(512, torch.Generator().manual_seed(7))
        with torch.no_grad():
            arrays['eval_x'] = x_eval
            arrays['eval_logp_before'] = log_density(x_eval)
        # one gradient before training, for the backward kernels alone (first batch of the run)
        gen = torch.Generator().manual_seed(2024)
        opt = torch.optim.Adam(layers.parameters(), lr=lr)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, steps, 0)
        losses = []
        for step in range(steps):                      # model.py:233-244 train_step
            x = nf_data.two_moons(B, gen)
            opt.zero_grad()
            loss = -log_density(x).mean()
            loss.backward()
            if step == 0:
                for j, f in enumerate(layers):
                    lin = [m for m in f.made.model if isinstance(m, made.SimpleMaskedLinear)]
                    for i, m in enumerate(lin):
                        arrays['l%d/gw%d_step0' % (j, i)] = m.weight.grad.clone()
                        arrays['l%d/gb%d_step0' % (j, i)] = m.bias.grad.clone()
            opt.step()
            sched.step()
            losses.append(loss.item())
        arrays['loss'] = np.array(losses)
        for j, f in enumerate(layers):
            lin = [m for m in f.made.model if isinstance(m, made.SimpleMaskedLinear)]
            for i, m in enumerate(lin):
                arrays['l%d/w%d_final' % (j, i)] = m.weight.detach()
                arrays['l%d/b%d_final' % (j, i)] = m.bias.detach()
        with torch.no_grad():
            arrays['eval_logp_after'] = log_density(x_eval)
            # the other direction on the trained model (model.py:87-92 from prior samples)
            z = torch.randn(256, D, generator=torch.Generator().manual_seed(9))
            h, logq = z, prior.log_prob(z)
            for layer in layers:
                h, logJ = layer.forward(h)
                logq = logq - logJ.squeeze()
            arrays.update(flow_z=z, flow_x=h, flow_logq=logq)
        print('maf training: loss %.6f -> %.6f' % (losses[0], losses[-1]))
        save('maf_train_moons', **arrays)
    finally:
        torch.set_default_dtype(torch.float32)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'maf_training':
        gen_maf_training()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'cubic_unconditional':
        gen_cubic_unconditional()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'random_mask':
        gen_random_mask()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'activations':
        gen_activations()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'unconditional':
        gen_unconditional_coupling()
        sys.exit(0)
    gen_spline()
    gen_made()
    gen_rq_ar()
    gen_affine_coupling()
    gen_maf()
    gen_maf_training()
    gen_unconditional_coupling()
    gen_activations()
    gen_cubic_unconditional()
    gen_random_mask()
