"""CPU oracle for the flow-transform hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (plain PyTorch CPU ops, functional style, driven by a
``state_dict``) of the reference algorithm for the path named by BASELINE.json's
north_star.  It is the checker for ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; nothing in the product package
(``normalizing-flows_amd/``) imports it and it never imports the reference.

Parity status: PINNED.  ``tests/golden/gen_golden.py`` imported the reference in the
build container and wrote the fixtures under ``tests/golden/``; ``tests/test_oracle.py``
checks every function here against them.

It deliberately keeps the reference's *operation sequence* (per-call ``W * mask``,
boolean-mask compaction around the spline, the in-place ``+1e-6`` on the last knot) so
that timing it on host cores has the reference's performance character.

Reference lines followed (all under /root/reference):
  tile / sum_except_batch / searchsorted .......... helper_functions.py:382-398, 507-512
  rq_spline / rq_spline_linear_tails .............. models/splines/rational_quadratic.py:13-149
  made_masks / made_forward ....................... models/made.py:91-152, 196-207, 268-284, 352-359
  simple_made_forward ............................. models/made.py:11-89
  rq_autoregressive / affine_autoregressive ....... models/transforms.py:488-500, 540-566, 819-860
  coupling (rq / affine) .......................... models/transforms.py:80-135, 156-168, 212-247, 328-347
  rq_cdf (unconditional transform) ................ models/transforms.py:249-318
  residual_net_forward ............................ models/nn/resnet.py:38-57, 90-100
  maf_forward / maf_reverse ....................... models/flows.py:239-263
  legacy_flow_to_samples / legacy_reverse_to_samples models/model.py:87-108
  composite ....................................... models/transforms.py:21-36
  log_prob / sample ............................... models/model.py:78-108, models/distributions.py:105-107, 154-196
"""
import math

import numpy as np
import torch
from torch.nn import functional as F

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------- utils
def tile(x, n):
    """[a,b,c] -> [a,a,..,b,b,..,c,c,..] (n copies each). helper_functions.py:382-390."""
    return x.reshape(-1, 1).expand(-1, n).reshape(-1)


def sum_except_batch(x):
    """helper_functions.py:393-398."""
    return x.reshape(x.shape[0], -1).sum(dim=1) if x.dim() > 1 else x


def searchsorted(knots, inputs, eps=1e-6):
    """Bin search; mutates the caller's last knot like helper_functions.py:507-512."""
    knots[..., -1] += eps
    return (inputs[..., None] >= knots).sum(dim=-1) - 1


# --------------------------------------------------------------------------- spline
def _knots(unnorm, min_size, lo, hi):
    """softmax -> floor -> cumsum -> pad -> rescale -> pin ends -> re-difference.
    rational_quadratic.py:71-78 (widths) and :82-89 (heights)."""
    nb = unnorm.shape[-1]
    sizes = F.softmax(unnorm, dim=-1)
    sizes = min_size + (1 - min_size * nb) * sizes
    cum = torch.cumsum(sizes, dim=-1)
    cum = F.pad(cum, pad=(1, 0), mode='constant', value=0.0)
    cum = (hi - lo) * cum + lo
    cum[..., 0] = lo
    cum[..., -1] = hi
    return cum, cum[..., 1:] - cum[..., :-1]


def rq_spline(inputs, uw, uh, ud, inverse=False, left=0., right=1., bottom=0., top=1.,
              min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3,
              return_bins=False):
    """Monotone rational-quadratic spline on [left,right]->[bottom,top].
    rational_quadratic.py:54-149.  ``ud`` has K+1 entries."""
    nb = uw.shape[-1]
    if min_bin_width * nb > 1.0:
        raise ValueError('Minimal bin width too large for the number of bins')
    if min_bin_height * nb > 1.0:
        raise ValueError('Minimal bin height too large for the number of bins')

    cumw, w = _knots(uw, min_bin_width, left, right)
    d = min_derivative + F.softplus(ud)
    cumh, h = _knots(uh, min_bin_height, bottom, top)

    idx = searchsorted(cumh if inverse else cumw, inputs)[..., None]

    def pick(t):
        return t.gather(-1, idx)[..., 0]

    x0 = pick(cumw)
    bw = pick(w)
    y0 = pick(cumh)
    delta = h / w
    s = pick(delta)
    d0 = pick(d)
    d1 = pick(d[..., 1:])
    bh = pick(h)

    if inverse:
        dy = inputs - y0
        curv = d0 + d1 - 2 * s
        a = dy * curv + bh * (s - d0)
        b = bh * d0 - dy * curv
        c = -s * dy
        disc = b.pow(2) - 4 * a * c
        assert (disc >= 0).all()
        root = (2 * c) / (-b - torch.sqrt(disc))
        outputs = root * bw + x0
        t1mt = root * (1 - root)
        den = s + curv * t1mt
        dnum = s.pow(2) * (d1 * root.pow(2) + 2 * s * t1mt + d0 * (1 - root).pow(2))
        lad = -(torch.log(dnum) - 2 * torch.log(den))
    else:
        theta = (inputs - x0) / bw
        t1mt = theta * (1 - theta)
        num = bh * (s * theta.pow(2) + d0 * t1mt)
        den = s + (d0 + d1 - 2 * s) * t1mt
        outputs = y0 + num / den
        dnum = s.pow(2) * (d1 * theta.pow(2) + 2 * s * t1mt + d0 * (1 - theta).pow(2))
        lad = torch.log(dnum) - 2 * torch.log(den)
    if return_bins:
        return outputs, lad, idx[..., 0]
    return outputs, lad


def rq_spline_linear_tails(inputs, uw, uh, ud, inverse=False, tail_bound=1.,
                           min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3,
                           return_bins=False):
    """Identity outside [-T,T], spline inside, via boolean-mask gather/scatter.
    rational_quadratic.py:13-52.  ``ud`` has K-1 entries (boundary slopes pinned to 1)."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    outside = ~inside
    outputs = torch.zeros_like(inputs)
    lad = torch.zeros_like(inputs)

    ud = F.pad(ud, pad=(1, 1))
    const = np.log(np.exp(1 - min_derivative) - 1)
    ud[..., 0] = const
    ud[..., -1] = const

    outputs[outside] = inputs[outside]
    lad[outside] = 0

    res = rq_spline(inputs[inside], uw[inside, :], uh[inside, :], ud[inside, :],
                    inverse=inverse, left=-tail_bound, right=tail_bound,
                    bottom=-tail_bound, top=tail_bound, min_bin_width=min_bin_width,
                    min_bin_height=min_bin_height, min_derivative=min_derivative,
                    return_bins=return_bins)
    outputs[inside], lad[inside] = res[0], res[1]
    if return_bins:
        bins = torch.full(inputs.shape, -1, dtype=torch.int64)
        bins[inside] = res[2]
        return outputs, lad, bins
    return outputs, lad


# --------------------------------------------------------------------------- MADE
def made_masks(features, hidden_features, num_blocks=2, output_multiplier=1,
               use_residual_blocks=True):
    """Degrees and 0/1 masks of the non-random MADE.  made.py:91-93, 119-148, 294-350.
    Returns dict name -> (mask float32 [out,in], degrees int64 [out]) keyed like the
    reference state_dict ('initial_layer', 'blocks.0.linear_layers.0', ..., 'final_layer')."""
    out = {}
    in_deg = torch.arange(1, features + 1)

    def hidden(prev):
        hi = max(1, features - 1)
        lo = min(1, features - 1)
        deg = torch.arange(hidden_features) % hi + lo
        return (deg[:, None] >= prev).float(), deg

    m, deg = hidden(in_deg)
    out['initial_layer'] = (m, deg)
    prev = deg
    for i in range(num_blocks):
        if use_residual_blocks:
            m0, d0 = hidden(prev)
            m1, d1 = hidden(d0)
            out['blocks.%d.linear_layers.0' % i] = (m0, d0)
            out['blocks.%d.linear_layers.1' % i] = (m1, d1)
            prev = d1
        else:
            m0, d0 = hidden(prev)
            out['blocks.%d.linear' % i] = (m0, d0)
            prev = d0
    odeg = tile(torch.arange(1, features + 1), output_multiplier)
    out['final_layer'] = ((odeg[:, None] > prev).float(), odeg)
    return out


def _masked_linear(sd, name, x):
    """F.linear(x, W*mask, b) with the mask multiplied on every call.  made.py:150-152."""
    return F.linear(x, sd[name + '.weight'] * sd[name + '.mask'], sd[name + '.bias'])


def _bn_eval(sd, name, x, eps=1e-3):
    """Eval-mode BatchNorm1d(eps=1e-3).  made.py:232-236 / resnet.py:22-26."""
    return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'],
                        sd[name + '.weight'], sd[name + '.bias'], False, 0.0, eps)


def made_forward(sd, x, prefix='', context=None, activation=F.relu):
    """MADE.forward in eval mode (dropout off, BN uses running stats); `activation` as the constructor argument.
    made.py:352-359 with blocks :196-207 / :268-284.  Block kind, BN and context layers
    are discovered from the state_dict keys."""
    p = prefix
    h = _masked_linear(sd, p + 'initial_layer', x)
    if context is not None:
        h = h + F.linear(context, sd[p + 'context_layer.weight'], sd[p + 'context_layer.bias'])
    i = 0
    while True:
        b = '%sblocks.%d.' % (p, i)
        if b + 'linear_layers.0.weight' in sd:
            t = h
            bn = b + 'batch_norm_layers.0.weight' in sd
            if bn:
                t = _bn_eval(sd, b + 'batch_norm_layers.0', t)
            t = activation(t)
            t = _masked_linear(sd, b + 'linear_layers.0', t)
            if bn:
                t = _bn_eval(sd, b + 'batch_norm_layers.1', t)
            t = activation(t)
            t = _masked_linear(sd, b + 'linear_layers.1', t)
            if context is not None:
                c = F.linear(context, sd[b + 'context_layer.weight'], sd[b + 'context_layer.bias'])
                t = F.glu(torch.cat((t, c), dim=1), dim=1)
            h = h + t
        elif b + 'linear.weight' in sd:
            t = h
            if b + 'batch_norm.weight' in sd:
                t = _bn_eval(sd, b + 'batch_norm', t)
            h = activation(_masked_linear(sd, b + 'linear', t))
        else:
            break
        i += 1
    return _masked_linear(sd, p + 'final_layer', h)


def simple_made_forward(weights, biases, masks, x, gaussian=True):
    """Linear-ReLU stack with mask*W; sigmoid head when not gaussian.  made.py:20-21, 50-55."""
    h = x
    n = len(weights)
    for i in range(n):
        h = F.linear(h.clone(), masks[i] * weights[i], biases[i])
        if i < n - 1:
            h = F.relu(h)
    return h if gaussian else torch.sigmoid(h)


# --------------------------------------------------------------------------- autoregressive
def _rq_ar_elementwise(x, params, num_bins, tail_bound, inverse, mins, return_bins=False):
    """view(B,D,P), slice w/h/d, spline, per-sample sum.  transforms.py:819-860.
    (MADE has no `hidden_features` attribute, so there is no 1/sqrt(H) scaling.)"""
    B, D = x.shape
    tp = params.view(B, D, -1)
    uw = tp[..., :num_bins]
    uh = tp[..., num_bins:2 * num_bins]
    ud = tp[..., 2 * num_bins:]
    res = rq_spline_linear_tails(x, uw, uh, ud, inverse=inverse, tail_bound=tail_bound,
                                 min_bin_width=mins[0], min_bin_height=mins[1],
                                 min_derivative=mins[2], return_bins=return_bins)
    if return_bins:
        return res[0], sum_except_batch(res[1]), res[2]
    return res[0], sum_except_batch(res[1])


def rq_autoregressive_forward(sd, x, num_bins, tail_bound, prefix='autoregressive_net.',
                              mins=(1e-3, 1e-3, 1e-3), return_bins=False, context=None, activation=F.relu):
    """One conditioner pass + elementwise forward.  transforms.py:488-491."""
    params = made_forward(sd, x, prefix, context, activation)
    return _rq_ar_elementwise(x, params, num_bins, tail_bound, False, mins, return_bins)


def rq_autoregressive_inverse(sd, z, num_bins, tail_bound, prefix='autoregressive_net.',
                              mins=(1e-3, 1e-3, 1e-3), context=None, activation=F.relu):
    """D sequential full passes starting from zeros.  transforms.py:493-500."""
    out = torch.zeros_like(z)
    lad = None
    for _ in range(int(np.prod(z.shape[1:]))):
        params = made_forward(sd, out, prefix, context, activation)
        out, lad = _rq_ar_elementwise(z, params, num_bins, tail_bound, True, mins)
    return out, lad


def _affine_scale_shift_interleaved(params, D):
    p = params.view(-1, D, 2)
    return torch.sigmoid(p[..., 0] + 2.) + 1e-3, p[..., 1]


def affine_autoregressive_forward(sd, x, prefix='autoregressive_net.', context=None):
    """transforms.py:540-546, 556-566."""
    scale, shift = _affine_scale_shift_interleaved(made_forward(sd, x, prefix, context), x.shape[1])
    return scale * x + shift, sum_except_batch(torch.log(scale))


def affine_autoregressive_inverse(sd, z, prefix='autoregressive_net.', context=None):
    """transforms.py:493-500 with :548-554."""
    out = torch.zeros_like(z)
    lad = None
    for _ in range(z.shape[1]):
        scale, shift = _affine_scale_shift_interleaved(made_forward(sd, out, prefix, context), z.shape[1])
        out = (z - shift) / scale
        lad = -sum_except_batch(torch.log(scale))
    return out, lad


# --------------------------------------------------------------------------- coupling
def residual_net_forward(sd, x, prefix='transform_net.', context=None, activation=F.relu):
    """Dense residual MLP in eval mode.  resnet.py:90-100, block :38-57."""
    p = prefix
    if context is not None:
        x = torch.cat((x, context), dim=1)
    h = F.linear(x, sd[p + 'initial_layer.weight'], sd[p + 'initial_layer.bias'])
    i = 0
    while '%sblocks.%d.linear_layers.0.weight' % (p, i) in sd:
        b = '%sblocks.%d.' % (p, i)
        bn = b + 'batch_norm_layers.0.weight' in sd
        t = h
        if bn:
            t = _bn_eval(sd, b + 'batch_norm_layers.0', t)
        t = activation(t)
        t = F.linear(t, sd[b + 'linear_layers.0.weight'], sd[b + 'linear_layers.0.bias'])
        if bn:
            t = _bn_eval(sd, b + 'batch_norm_layers.1', t)
        t = activation(t)
        t = F.linear(t, sd[b + 'linear_layers.1.weight'], sd[b + 'linear_layers.1.bias'])
        if context is not None:
            c = F.linear(context, sd[b + 'context_layer.weight'], sd[b + 'context_layer.bias'])
            t = F.glu(torch.cat((t, c), dim=1), dim=1)
        h = h + t
        i += 1
    return F.linear(h, sd[p + 'final_layer.weight'], sd[p + 'final_layer.bias'])


def rq_cdf(sd, x, tails, tail_bound, inverse=False, mins=(1e-3, 1e-3, 1e-3), prefix='unconditional_transform.'):
    """PiecewiseRationalQuadraticCDF: learned parameters shared across the batch.  transforms.py:284-318."""
    b = x.shape[0]
    uw = sd[prefix + 'unnormalized_widths'][None, ...].expand(b, -1, -1)
    uh = sd[prefix + 'unnormalized_heights'][None, ...].expand(b, -1, -1)
    ud = sd[prefix + 'unnormalized_derivatives'][None, ...].expand(b, -1, -1)
    if tails is None:
        y, ld = rq_spline(x, uw, uh, ud, inverse=inverse, min_bin_width=mins[0], min_bin_height=mins[1],
                          min_derivative=mins[2])
    else:
        y, ld = rq_spline_linear_tails(x, uw, uh, ud, inverse=inverse, tail_bound=tail_bound, min_bin_width=mins[0],
                                       min_bin_height=mins[1], min_derivative=mins[2])
    return y, sum_except_batch(ld)


def cubic_cdf(sd, x, tails, tail_bound, inverse=False, mins=(1e-3, 1e-3), prefix='unconditional_transform.'):
    """PiecewiseCubicCDF: learned parameters shared across the batch.  transforms.py:417-473 (with tails=None the
    reference calls the dangling `splines.cubic_spline`; read as cubic.py:56-221 on its default interval)."""
    b = x.shape[0]
    uw = sd[prefix + 'unnormalized_widths'][None, ...].expand(b, -1, -1)
    uh = sd[prefix + 'unnormalized_heights'][None, ...].expand(b, -1, -1)
    ul = sd[prefix + 'unnorm_derivatives_left'][None, ...].expand(b, -1, -1)
    ur = sd[prefix + 'unnorm_derivatives_right'][None, ...].expand(b, -1, -1)
    if tails is None:
        y, ld = cubic_spline(x, uw, uh, ul, ur, inverse=inverse, min_bin_width=mins[0], min_bin_height=mins[1])
    else:
        y, ld = unconstrained_cubic_spline(x, uw, uh, ul, ur, inverse=inverse, tail_bound=tail_bound,
                                           min_bin_width=mins[0], min_bin_height=mins[1])
    return y, sum_except_batch(ld)


def cubic_coupling(sd, x, num_bins, tails, tail_bound, hidden_features, inverse=False, mins=(1e-3, 1e-3),
                   unconditional=False):
    """Cubic coupling layer, ResidualNet conditioner.  transforms.py:349-415 (logits / sqrt(hidden_features))."""
    def elementwise(xt, params):
        tp = params.reshape(xt.shape[0], xt.shape[1], -1)
        uw = tp[..., :num_bins] / np.sqrt(hidden_features)
        uh = tp[..., num_bins:2 * num_bins] / np.sqrt(hidden_features)
        ul, ur = tp[..., 2 * num_bins:2 * num_bins + 1], tp[..., 2 * num_bins + 1:]
        if tails is None:
            yy, ll = cubic_spline(xt, uw, uh, ul, ur, inverse=inverse, min_bin_width=mins[0], min_bin_height=mins[1])
        else:
            yy, ll = unconstrained_cubic_spline(xt, uw, uh, ul, ur, inverse=inverse, tail_bound=tail_bound,
                                                min_bin_width=mins[0], min_bin_height=mins[1])
        return yy, sum_except_batch(ll)

    unc = (lambda v, inv: cubic_cdf(sd, v, tails, tail_bound, inverse=inv, mins=mins)) if unconditional else None
    return _coupling(sd, x, elementwise, unconditional=unc, inverse=inverse)


def _coupling(sd, x, elementwise, unconditional=None, inverse=False):
    """Split by index buffers, conditioner on identity half, scatter.  transforms.py:80-135.
    unconditional(ident, inverse) -> (ident', logabsdet): the identity half's own transform (:98-101 after the
    conditioner has read the original half in forward; :120-123 before the conditioner in inverse)."""
    idf = sd['identity_features']
    trf = sd['transform_features']
    ident = x[:, idf]
    lad_id = None
    if unconditional is not None and inverse:
        ident, lad_id = unconditional(ident, True)
    params = residual_net_forward(sd, ident)
    t, lad = elementwise(x[:, trf], params)
    if unconditional is not None and not inverse:
        ident, lad_id = unconditional(ident, False)
    if lad_id is not None:
        lad = lad + lad_id
    out = torch.empty_like(x)
    out[:, idf] = ident
    out[:, trf] = t
    return out, lad


def rq_coupling(sd, x, num_bins, tails, tail_bound, hidden_features, inverse=False,
                mins=(1e-3, 1e-3, 1e-3), return_bins=False, unconditional=False):
    """RQ coupling layer, ResidualNet conditioner.  transforms.py:156-168, 212-247.
    widths/heights are divided by sqrt(hidden_features) (ResidualNet sets the attribute)."""
    bins_box = []

    def elementwise(xt, params):
        b, d = xt.shape
        tp = params.reshape(b, d, -1)
        uw = tp[..., :num_bins]
        uh = tp[..., num_bins:2 * num_bins]
        ud = tp[..., 2 * num_bins:]
        uw /= np.sqrt(hidden_features)
        uh /= np.sqrt(hidden_features)
        if tails is None:
            res = rq_spline(xt, uw, uh, ud, inverse=inverse, min_bin_width=mins[0],
                            min_bin_height=mins[1], min_derivative=mins[2],
                            return_bins=return_bins)
        else:
            res = rq_spline_linear_tails(xt, uw, uh, ud, inverse=inverse, tail_bound=tail_bound,
                                         min_bin_width=mins[0], min_bin_height=mins[1],
                                         min_derivative=mins[2], return_bins=return_bins)
        if return_bins:
            bins_box.append(res[2])
        return res[0], sum_except_batch(res[1])

    unc = (lambda v, inv: rq_cdf(sd, v, tails, tail_bound, inverse=inv, mins=mins)) if unconditional else None
    out, lad = _coupling(sd, x, elementwise, unconditional=unc, inverse=inverse)
    if return_bins:
        return out, lad, bins_box[0]
    return out, lad


def affine_coupling(sd, x, inverse=False):
    """Blocked params: shift = p[:, :d], scale = sigmoid(p[:, d:]+2)+1e-3.  transforms.py:328-347."""
    def elementwise(xt, params):
        d = xt.shape[1]
        shift = params[:, :d]
        scale = torch.sigmoid(params[:, d:] + 2) + 1e-3
        ls = sum_except_batch(torch.log(scale))
        if inverse:
            return (xt - shift) / scale, -ls
        return xt * scale + shift, ls
    return _coupling(sd, x, elementwise)


# --------------------------------------------------------------------------- legacy MAF
def maf_forward(weights, biases, masks, z, reverse_input=True):
    """x=(z-mu)*exp(logp/2), optional flip, logJ=sum(logp)/2, in float64.  flows.py:239-249."""
    out = simple_made_forward(weights, biases, masks, z.double())
    mu, logp = torch.chunk(out, 2, dim=1)
    x = (z - mu) * torch.exp(0.5 * logp)
    if reverse_input:
        x = x.flip(dims=(1,))
    return x, 0.5 * torch.sum(logp, dim=1)


def maf_reverse(weights, biases, masks, x, reverse_input=True):
    """D-pass loop; logJ comes from the LAST pass only.  flows.py:251-263."""
    if reverse_input:
        x = x.flip(dims=(1,))
    z = torch.zeros_like(x)
    mod = None
    for dim in range(x.shape[1]):
        out = simple_made_forward(weights, biases, masks, z)
        mu, logp = torch.chunk(out, 2, dim=1)
        mod = torch.clamp(-0.5 * logp, max=10)
        z[:, dim] = mu[:, dim] + x[:, dim] * torch.exp(mod[:, dim])
    return z, torch.sum(mod, dim=1)


def legacy_flow_to_samples(layers, x, logq):
    """model.py:87-92 over legacy layers: `layers` = [(weights, biases, masks, reverse_input)]."""
    for w, b, m, ri in layers:
        x, logJ = maf_forward(w, b, m, x, ri)
        logq = logq - logJ.squeeze()
    return x, logq


def legacy_reverse_to_samples(layers, x, logq):
    """model.py:103-108 with a standard-normal prior (distributions.py:105-107): the log-density the stack assigns
    to data x; cfg-1 trains on `-mean` of it."""
    for w, b, m, ri in reversed(list(layers)):
        x, logJ = maf_reverse(w, b, m, x, ri)
        logq = logq + logJ.squeeze()
    return x, std_normal_log_prob(x) + logq


# --------------------------------------------------------------------------- stacks
def composite(funcs, x):
    """Run ``funcs`` in order, summing logabsdet.  transforms.py:21-28 (scratch allocated
    like the input instead of in the default dtype)."""
    total = torch.zeros(x.shape[0], dtype=x.dtype)
    for f in funcs:
        x, lad = f(x)
        total = total + lad
    return x, total


def std_normal_log_prob(x):
    """N(0,I) log-density summed over features.  distributions.py:105-107."""
    return (-0.5 * x.pow(2) - 0.5 * LOG_2PI).reshape(x.shape[0], -1).sum(dim=1)


def flow_log_prob(funcs, x):
    """log p(x) = base.log_prob(T(x)) + logabsdet.  model.py:103-108, distributions.py:154-157."""
    z, lad = composite(funcs, x)
    return std_normal_log_prob(z) + lad


def split_state_dict(sd, j):
    """Sub-dict of a CompositeTransform state_dict for layer j (prefix `_transforms.j.`)."""
    pre = '_transforms.%d.' % j
    return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}


# --------------------------------------------------------------------------- synthetic weights
def seeded_made_state_dict(features, hidden_features, output_multiplier, num_blocks=2, seed=0,
                           scale=None, use_residual_blocks=True, dtype=torch.float32,
                           prefix='autoregressive_net.'):
    """Deterministic MADE weights (CPU generator => identical on every machine), in the
    reference's state_dict layout.  Initialisation statistics mirror nn.Linear's
    Kaiming-uniform and the U(-1e-3,1e-3) last layer of each residual block
    (made.py:264-266); ``scale`` overrides with N(0, scale^2) everywhere."""
    g = torch.Generator().manual_seed(seed)
    masks = made_masks(features, hidden_features, num_blocks, output_multiplier, use_residual_blocks)
    sd = {}
    for name, (m, deg) in masks.items():
        out_f, in_f = m.shape
        if scale is not None:
            w = torch.randn(out_f, in_f, generator=g, dtype=torch.float64) * scale
            b = torch.randn(out_f, generator=g, dtype=torch.float64) * scale
        else:
            bound = 1e-3 if name.endswith('linear_layers.1') else 1.0 / math.sqrt(in_f)
            w = (torch.rand(out_f, in_f, generator=g, dtype=torch.float64) * 2 - 1) * bound
            b = (torch.rand(out_f, generator=g, dtype=torch.float64) * 2 - 1) * bound
        sd[prefix + name + '.weight'] = w.to(dtype)
        sd[prefix + name + '.bias'] = b.to(dtype)
        sd[prefix + name + '.mask'] = m.to(dtype)
        sd[prefix + name + '.degrees'] = deg
    return sd


def compute_ess(logp, logq):
    """helper_functions.py:36-40: exp(2 lse(logw) - lse(2 logw)) / N."""
    logw = logp - logq
    log_ess = 2 * torch.logsumexp(logw, dim=0) - torch.logsumexp(2 * logw, dim=0)
    return torch.exp(log_ess) / len(logw)


def forward_kl(logp, logq):
    """model.py:126-130."""
    return (logp - logq).mean()


def reverse_kl(logq, logp):
    """model.py:132-136."""
    return (logq - logp).mean()


# --------------------------------------------------------------------------- linear / quadratic splines
def linear_spline(inputs, unnormalized_pdf, inverse=False, left=0., right=1., bottom=0., top=1.,
                  return_bins=False):
    """Piecewise-linear CDF on K equal bins (models/splines/linear.py:35-94).  Forward: bin = floor(x K);
    inverse: search over the cdf (whose last entry the search raises by 1e-6 in place, :52 + the slopes
    of :58-60 are then built from the raised value)."""
    inputs = (inputs - bottom) / (top - bottom) if inverse else (inputs - left) / (right - left)
    nb = unnormalized_pdf.size(-1)
    pdf = F.softmax(unnormalized_pdf, dim=-1)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf[..., -1] = 1.
    cdf = F.pad(cdf, pad=(1, 0), mode='constant', value=0.0)
    if inverse:
        idx = searchsorted(cdf, inputs)
        edges = torch.linspace(0, 1, nb + 1).view([1] * inputs.dim() + [-1]).expand(*inputs.shape, -1)
        slopes = (cdf[..., 1:] - cdf[..., :-1]) / (edges[..., 1:] - edges[..., :-1])
        offsets = cdf[..., 1:] - slopes * edges[..., 1:]
        sel = idx.unsqueeze(-1)
        s_in = slopes.gather(-1, sel)[..., 0]
        o_in = offsets.gather(-1, sel)[..., 0]
        outputs = torch.clamp((inputs - o_in) / s_in, 0, 1)
        lad = -torch.log(s_in)
    else:
        pos = inputs * nb
        idx = torch.floor(pos).long()
        idx[idx >= nb] = nb - 1
        alpha = pos - idx.float()
        p_in = pdf.gather(-1, idx[..., None])[..., 0]
        outputs = cdf.gather(-1, idx[..., None])[..., 0]
        outputs += alpha * p_in
        outputs = torch.clamp(outputs, 0, 1)
        lad = torch.log(p_in) - np.log(1.0 / nb)
    if inverse:
        outputs = outputs * (right - left) + left
        lad = lad - math.log(top - bottom) + math.log(right - left)
    else:
        outputs = outputs * (top - bottom) + bottom
        lad = lad + math.log(top - bottom) - math.log(right - left)
    if return_bins:
        return outputs, lad, idx
    return outputs, lad


def unconstrained_linear_spline(inputs, unnormalized_pdf, inverse=False, tail_bound=1., return_bins=False):
    """Identity outside [-T, T] (models/splines/linear.py:9-33)."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    outputs = torch.zeros_like(inputs)
    lad = torch.zeros_like(inputs)
    outputs[~inside] = inputs[~inside]
    res = linear_spline(inputs[inside], unnormalized_pdf[inside, :], inverse=inverse, left=-tail_bound,
                        right=tail_bound, bottom=-tail_bound, top=tail_bound, return_bins=return_bins)
    outputs[inside], lad[inside] = res[0], res[1]
    if return_bins:
        bins = torch.full(inputs.shape, -1, dtype=torch.int64)
        bins[inside] = res[2]
        return outputs, lad, bins
    return outputs, lad


def quadratic_spline(inputs, uw, uh, inverse=False, left=0., right=1., bottom=0., top=1.,
                     min_bin_width=1e-3, min_bin_height=1e-3, return_bins=False):
    """Piecewise-quadratic CDF = integral of a piecewise-linear density (models/splines/quadratic.py:48-135).
    `uh` has K+1 entries, or K-1 with the two boundary heights chosen so that they normalise to 1 (:73-84)."""
    inputs = (inputs - bottom) / (top - bottom) if inverse else (inputs - left) / (right - left)
    nb = uw.shape[-1]
    if min_bin_width * nb > 1.0:
        raise ValueError('Minimal bin width too large for the number of bins')
    if min_bin_height * nb > 1.0:
        raise ValueError('Minimal bin height too large for the number of bins')
    widths = F.softmax(uw, dim=-1)
    widths = min_bin_width + (1 - min_bin_width * nb) * widths
    he = torch.exp(uh)
    if he.shape[-1] == nb - 1:
        first = 0.5 * widths[..., 0]
        last = 0.5 * widths[..., -1]
        num = (0.5 * first * he[..., 0] + 0.5 * last * he[..., -1]
               + torch.sum(((he[..., :-1] + he[..., 1:]) / 2) * widths[..., 1:-1], dim=-1))
        const = (num / (1 - 0.5 * first - 0.5 * last))[..., None]
        he = torch.cat([const, he, const], dim=-1)
    area = torch.sum(((he[..., :-1] + he[..., 1:]) / 2) * widths, dim=-1)[..., None]
    heights = he / area
    heights = min_bin_height + (1 - min_bin_height) * heights
    left_cdf = torch.cumsum(((heights[..., :-1] + heights[..., 1:]) / 2) * widths, dim=-1)
    left_cdf[..., -1] = 1.
    left_cdf = F.pad(left_cdf, pad=(1, 0), mode='constant', value=0.0)
    locs = torch.cumsum(widths, dim=-1)
    locs[..., -1] = 1.
    locs = F.pad(locs, pad=(1, 0), mode='constant', value=0.0)
    idx = (searchsorted(left_cdf, inputs) if inverse else searchsorted(locs, inputs))[..., None]
    loc_in = locs.gather(-1, idx)[..., 0]
    w_in = widths.gather(-1, idx)[..., 0]
    c = left_cdf.gather(-1, idx)[..., 0]
    hl = heights.gather(-1, idx)[..., 0]
    hr = heights.gather(-1, idx + 1)[..., 0]
    a = 0.5 * (hr - hl) * w_in
    b = hl * w_in
    if inverse:
        c_ = c - inputs
        alpha = (-b + torch.sqrt(b.pow(2) - 4 * a * c_)) / (2 * a)
        outputs = torch.clamp(alpha * w_in + loc_in, 0, 1)
        lad = -torch.log(alpha * (hr - hl) + hl)
    else:
        alpha = (inputs - loc_in) / w_in
        outputs = torch.clamp(a * alpha.pow(2) + b * alpha + c, 0, 1)
        lad = torch.log(alpha * (hr - hl) + hl)
    if inverse:
        outputs = outputs * (right - left) + left
        lad = lad - math.log(top - bottom) + math.log(right - left)
    else:
        outputs = outputs * (top - bottom) + bottom
        lad = lad + math.log(top - bottom) - math.log(right - left)
    if return_bins:
        return outputs, lad, idx[..., 0]
    return outputs, lad


def unconstrained_quadratic_spline(inputs, uw, uh, inverse=False, tail_bound=1., min_bin_width=1e-3,
                                   min_bin_height=1e-3, return_bins=False):
    """Identity outside [-T, T]; K-1 heights (models/splines/quadratic.py:11-46)."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    outputs = torch.zeros_like(inputs)
    lad = torch.zeros_like(inputs)
    outputs[~inside] = inputs[~inside]
    assert uh.shape[-1] == uw.shape[-1] - 1
    res = quadratic_spline(inputs[inside], uw[inside, :], uh[inside, :], inverse=inverse, left=-tail_bound,
                           right=tail_bound, bottom=-tail_bound, top=tail_bound, min_bin_width=min_bin_width,
                           min_bin_height=min_bin_height, return_bins=return_bins)
    outputs[inside], lad[inside] = res[0], res[1]
    if return_bins:
        bins = torch.full(inputs.shape, -1, dtype=torch.int64)
        bins[inside] = res[2]
        return outputs, lad, bins
    return outputs, lad


def cbrt(x):
    """helper_functions.py:514-516."""
    return torch.sign(x) * torch.exp(torch.log(torch.abs(x)) / 3.0)


def cubic_spline(inputs, uw, uh, ul, ur, inverse=False, left=-10., right=10., bottom=-10., top=10.,
                 min_bin_width=1e-3, min_bin_height=1e-3, eps=1e-5, quadratic_threshold=1e-3, return_bins=False):
    """Monotone piecewise-cubic spline (Steffen-type knot derivatives), inverse by Cardano / the trigonometric
    form with the root picked by bin membership, near-quadratic bins re-solved (models/splines/cubic.py:56-221)."""
    nb = uw.shape[-1]
    if min_bin_width * nb > 1.0:
        raise ValueError('Minimal bin width too large for the number of bins')
    if min_bin_height * nb > 1.0:
        raise ValueError('Minimal bin height too large for the number of bins')
    inputs = (inputs - bottom) / (top - bottom) if inverse else (inputs - left) / (right - left)
    widths = F.softmax(uw, dim=-1)
    widths = min_bin_width + (1 - min_bin_width * nb) * widths
    cumw = torch.cumsum(widths, dim=-1)
    cumw[..., -1] = 1
    cumw = F.pad(cumw, pad=(1, 0), mode='constant', value=0.0)
    heights = F.softmax(uh, dim=-1)
    heights = min_bin_height + (1 - min_bin_height * nb) * heights
    cumh = torch.cumsum(heights, dim=-1)
    cumh[..., -1] = 1
    cumh = F.pad(cumh, pad=(1, 0), mode='constant', value=0.0)
    slopes = heights / widths
    m1 = torch.min(torch.abs(slopes[..., :-1]), torch.abs(slopes[..., 1:]))
    m2 = (0.5 * (widths[..., 1:] * slopes[..., :-1] + widths[..., :-1] * slopes[..., 1:])
          / (widths[..., :-1] + widths[..., 1:]))
    ms = torch.min(m1, m2)
    d_left = torch.sigmoid(ul) * 3 * slopes[..., 0][..., None]
    d_right = torch.sigmoid(ur) * 3 * slopes[..., -1][..., None]
    deriv = ms * (torch.sign(slopes[..., :-1]) + torch.sign(slopes[..., 1:]))
    deriv = torch.cat([d_left, deriv, d_right], dim=-1)
    a = (deriv[..., :-1] + deriv[..., 1:] - 2 * slopes) / widths.pow(2)
    b = (3 * slopes - 2 * deriv[..., :-1] - deriv[..., 1:]) / widths
    c = deriv[..., :-1]
    d = cumh[..., :-1]
    idx = (searchsorted(cumh, inputs) if inverse else searchsorted(cumw, inputs))[..., None]
    ia = a.gather(-1, idx)[..., 0]
    ib = b.gather(-1, idx)[..., 0]
    ic = c.gather(-1, idx)[..., 0]
    id_ = d.gather(-1, idx)[..., 0]
    lw = cumw.gather(-1, idx)[..., 0]
    rw = cumw.gather(-1, idx + 1)[..., 0]
    if inverse:
        b_ = (ib / ia) / 3.
        c_ = (ic / ia) / 3.
        d_ = (id_ - inputs) / ia
        delta_1 = -b_.pow(2) + c_
        delta_2 = -c_ * b_ + d_
        delta_3 = b_ * d_ - c_.pow(2)
        disc = 4. * delta_1 * delta_3 - delta_2.pow(2)
        dep_1 = -2. * b_ * delta_1 + delta_2
        dep_2 = delta_1
        three = disc >= 0
        one = disc < 0
        outputs = torch.zeros_like(inputs)
        p = cbrt((-dep_1[one] + torch.sqrt(-disc[one])) / 2.)
        q = cbrt((-dep_1[one] - torch.sqrt(-disc[one])) / 2.)
        outputs[one] = (p + q) - b_[one] + lw[one]
        theta = torch.atan2(torch.sqrt(disc[three]), -dep_1[three])
        theta /= 3.
        cr1 = torch.cos(theta)
        cr2 = torch.sin(theta)
        r1 = cr1
        r2 = -0.5 * cr1 - 0.5 * math.sqrt(3) * cr2
        r3 = -0.5 * cr1 + 0.5 * math.sqrt(3) * cr2
        scale = 2 * torch.sqrt(-dep_2[three])
        shift = -b_[three] + lw[three]
        r1 = r1 * scale + shift
        r2 = r2 * scale + shift
        r3 = r3 * scale + shift
        k1 = ((lw[three] - eps) < r1).float()
        k1 *= (r1 < (rw[three] + eps)).float()
        k2 = ((lw[three] - eps) < r2).float()
        k2 *= (r2 < (rw[three] + eps)).float()
        k3 = ((lw[three] - eps) < r3).float()
        k3 *= (r3 < (rw[three] + eps)).float()
        roots = torch.stack([r1, r2, r3], dim=-1)
        masks = torch.stack([k1, k2, k3], dim=-1)
        pick = torch.argsort(masks, dim=-1, descending=True)[..., 0][..., None]
        outputs[three] = torch.gather(roots, dim=-1, index=pick).view(-1)
        quad = ia.abs() < quadratic_threshold
        qa = ib[quad]
        qb = ic[quad]
        qc = id_[quad] - inputs[quad]
        alpha = (-qb + torch.sqrt(qb.pow(2) - 4 * qa * qc)) / (2 * qa)
        outputs[quad] = alpha + lw[quad]
        sh = outputs - lw
        lad = -torch.log(3 * ia * sh.pow(2) + 2 * ib * sh + ic)
    else:
        sh = inputs - lw
        outputs = ia * sh.pow(3) + ib * sh.pow(2) + ic * sh + id_
        lad = torch.log(3 * ia * sh.pow(2) + 2 * ib * sh + ic)
    if inverse:
        outputs = outputs * (right - left) + left
        lad = lad - math.log(top - bottom) + math.log(right - left)
    else:
        outputs = outputs * (top - bottom) + bottom
        lad = lad + math.log(top - bottom) - math.log(right - left)
    if return_bins:
        return outputs, lad, idx[..., 0]
    return outputs, lad


def unconstrained_cubic_spline(inputs, uw, uh, ul, ur, inverse=False, tail_bound=1., min_bin_width=1e-3,
                               min_bin_height=1e-3, return_bins=False):
    """Identity outside [-T, T] (models/splines/cubic.py:14-54)."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    outputs = torch.zeros_like(inputs)
    lad = torch.zeros_like(inputs)
    outputs[~inside] = inputs[~inside]
    res = cubic_spline(inputs[inside], uw[inside, :], uh[inside, :], ul[inside, :], ur[inside, :], inverse=inverse,
                       left=-tail_bound, right=tail_bound, bottom=-tail_bound, top=tail_bound,
                       min_bin_width=min_bin_width, min_bin_height=min_bin_height, return_bins=return_bins)
    outputs[inside], lad[inside] = res[0], res[1]
    if return_bins:
        bins = torch.full(inputs.shape, -1, dtype=torch.int64)
        bins[inside] = res[2]
        return outputs, lad, bins
    return outputs, lad
