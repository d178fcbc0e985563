"""torch.autograd glue for training by maximum likelihood (BASELINE.json config 5).

The reference trains by letting autograd differentiate its ATen op sequences (test2.py:127-135,
model.py:233-244).  Here every forward kernel has a hand-written backward kernel and these
``autograd.Function``s only connect them: nothing falls back to PyTorch math.

  MaskedLinearFn : y = act_in(x) @ (W*mask)^T + b (+ residual)
                   backward: grad_x = (grad_y @ (W*mask)) gated by x > 0   (nf_masked_linear_bwd_input)
                             grad_W = mask * grad_y^T @ act_in(x), grad_b  (nf_masked_linear_bwd_weight)
  RqsFn          : (y, logabsdet) = spline(x; params), forward direction, or the inverse direction differentiated
                   implicitly at its output (nf_rqs_inverse_bwd)
  AffineFn       : (y, logabsdet) = sigmoid-scale affine(x; params), either direction, AR or coupling layout
                   backward: recompute-from-inputs kernel nf_rqs_bwd -> grad_x, grad_params
  ResidualBlockFn : a plain MADE residual block as one node: the skip connection's gradient joins the first layer's
                   input-gradient GEMM in its epilogue (nf_masked_linear_split_bwd_input_add)
  StdNormalLogProbFn : log N(z; 0, I) + sign * logabsdet, one launch (nf_std_normal_log_prob)
  FinalRqsFn     : final MADE layer + RQ spline of a whole autoregressive layer as ONE node (f32 / bf16 modes): forward
                   nf_made_rqs_fwd_params (the parameters kept, not re-read between GEMM and spline); backward
                   nf_rqs_bwd_max + the final layer's two GEMMs - in bf16 mode with the parameter gradients as a bf16
                   plane between the three kernels (nf_rqs_bwd_max_bf16, nf_masked_linear_split_bwd_input_bf16,
                   nf_masked_linear_half_bwd_weight_bf16)
"""
import torch

from . import _lib


def _split_mode(x, width):
    """True when the matrix mode runs its hidden-layer GEMMs on the split-operand kernel (the split modes, and bf16
    as its one-term flavour) and that kernel takes this operand; then forward and input-gradient GEMMs run on it."""
    from . import made
    return made.get_matmul_dtype() in made._SPLIT_MODES and _lib.split_gemm_supported(x, width)


def _f16x2_mode():
    from . import made
    return made.get_matmul_dtype() == 'f16x2'


def _bf16_mode():
    from . import made
    return made.get_matmul_dtype() == 'bf16'


def _grad_sink(weight, bias, has_bias):
    """The pair (weight.grad, bias.grad) when this node may accumulate its weight gradient straight into them, else
    None.  parallel.GradientSynchronizer opts its parameters in (`_nf_grad_sink`): their .grad are persistent zeroed
    views of the flat buckets, so the weight-gradient kernels - which accumulate into their output - need neither a
    zero-filled temporary nor AccumulateGrad's add pass (4 small launches per layer and step, a 37 MB add for the final
    layer).  The node then returns None for these inputs; the parameters' post-accumulate hooks (the synchronizer's bucket
    countdown) still run once per backward: by the engine where it runs them for a None gradient, else by _sink_done."""
    def ok(p):
        g = getattr(p, 'grad', None)
        return (getattr(p, '_nf_grad_sink', False) and p.is_leaf and p.requires_grad and g is not None
                and g.dtype == torch.float32 and g.shape == p.shape and g.is_contiguous() and g.device == p.device
                and g.data_ptr() % 16 == 0 and not g.requires_grad and not p._backward_hooks)
    if not ok(weight) or (has_bias and not ok(bias)):
        return None
    return weight.grad, (bias.grad if has_bias else None)


_engine_fires_hooks = None


def _engine_fires_post_accumulate_hooks_without_a_gradient():
    """Does this torch run a leaf's post-accumulate-grad hooks when the node feeding it returned None?  (2.10 does;
    probed once on a 3-element CPU graph rather than assumed.)"""
    global _engine_fires_hooks
    if _engine_fires_hooks is None:
        class Probe(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                return x * 1.0

            @staticmethod
            def backward(ctx, g):
                return g, None
        w = torch.zeros(1, requires_grad=True)
        fired = []
        handle = w.register_post_accumulate_grad_hook(lambda p: fired.append(1))
        with torch.enable_grad():
            Probe.apply(torch.zeros(1, requires_grad=True), w).sum().backward()
        handle.remove()
        _engine_fires_hooks = bool(fired)
    return _engine_fires_hooks


def _sink_done(*params):
    """The parameters' post-accumulate-grad hooks after a direct accumulation - unless the engine runs them anyway."""
    if _engine_fires_post_accumulate_hooks_without_a_gradient():
        return
    for p in params:
        hooks = getattr(p, '_post_accumulate_grad_hooks', None) if p is not None else None
        if hooks:
            for h in list(hooks.values()):
                h(p)


class MaskedLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, layer, relu_in):
        ctx.x_max = None
        if _split_mode(x, layer.in_features):
            ctx.x_max = _lib.tagged_max(x)
            if ctx.x_max is None:
                ctx.x_max = _lib.maxabs(x)
            y = _lib.masked_linear_split(x, layer._packed_linear_split(), residual=residual, relu_in=relu_in,
                                         x_max=ctx.x_max)
        else:
            y = _lib.masked_linear(x, layer._packed_linear(), residual=residual, relu_in=relu_in)
        ctx.save_for_backward(x)
        ctx.layer = layer
        ctx.relu_in = relu_in
        ctx.has_residual = residual is not None
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        (x,) = ctx.saved_tensors
        grad_y = grad_y.contiguous()
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        sink = _grad_sink(*ctx.params, ctx.has_bias) if want_w else None
        gx, gw, gb = _linear_backward(ctx.layer, x, grad_y, ctx.relu_in, ctx.x_max, ctx.needs_input_grad[0], want_w,
                                      ctx.has_bias, into=sink)
        gres = grad_y if ctx.has_residual else None
        if sink is not None:
            _sink_done(*ctx.params)
            gw = gb = None
        return gx, gw, gb, gres, None, None


def _linear_backward(layer, x, grad_y, relu_in, x_max, want_x, want_w, has_bias, into=None, add=None):
    """(grad_x, grad_W, grad_b) of y = act_in(x) @ (W*mask)^T + b from grad_y, in the current matrix mode (`into`: the
    pair the weight gradient is accumulated into instead of fresh tensors, _grad_sink; `add`: a gradient that joins
    grad_x, in the input-gradient GEMM's epilogue where that kernel takes it)."""
    gx = gw = gb = None
    split_y = _split_mode(grad_y, layer.out_features)
    gy_max = _lib.tagged_max(grad_y)                   # left by the kernel that produced it (nf_rqs_bwd_max)
    if split_y and gy_max is None:
        gy_max = _lib.maxabs(grad_y)
    if want_x:
        if split_y:
            # dX = dY @ (W * mask): the same split-operand kernel on the transposed layer, ReLU gate in its epilogue
            fuse = add is not None and relu_in
            gx = _lib.masked_linear_split(grad_y, layer._packed_linear_split_t(), x_max=gy_max,
                                          gate=x if relu_in else None, reverse_k=True, add=add if fuse else None)
            if add is not None and not fuse:
                gx = gx + add
        else:
            gx = _lib.masked_linear_bwd_input(grad_y, layer._packed_linear_t(), gate=x if relu_in else None)
            if add is not None:
                gx = gx + add
    elif add is not None:
        gx = add
    if want_w:
        if _bf16_mode() and _lib.split_gemm_supported(grad_y, layer.out_features):
            # bf16 mode: one scaled fp16 term per operand (11 bits against bf16's 8, same MFMA rate)
            gw, gb = _lib.masked_linear_split_bwd_weight(x, grad_y, getattr(layer, 'mask', None),
                                                         layer._wgrad_ktab(), relu_in, want_bias=has_bias,
                                                         x_max=x_max, grad_y_max=gy_max, terms=1, into=into)
        elif split_y and x_max is not None and _f16x2_mode():
            gw, gb = _lib.masked_linear_split_bwd_weight(x, grad_y, getattr(layer, 'mask', None),
                                                         layer._wgrad_ktab(), relu_in,
                                                         want_bias=has_bias, x_max=x_max, grad_y_max=gy_max, into=into)
        else:
            gw, gb = _lib.masked_linear_bwd_weight(x, grad_y, getattr(layer, 'mask', None),
                                                   layer._wgrad_ktab(), relu_in, want_bias=has_bias, into=into)
    return gx, gw, gb


# One autograd node per plain residual block (False: two MaskedLinearFn nodes and autograd's own sum, A/B and tests)
FUSED_RESIDUAL_BLOCK = True


class ResidualBlockFn(torch.autograd.Function):
    """y = x + ML1(relu(ML0(relu(x)))) - a MADE residual block without BatchNorm, dropout or context (made.py:268-284) -
    as ONE node.  Forward: the two launches MaskedLinearFn makes.  Backward: the same four GEMMs, but the gradient that
    reaches x along the skip connection joins the first layer's input-gradient GEMM in its epilogue
    (nf_masked_linear_split_bwd_input_add) instead of in an add node of autograd's - which also left the sum without a
    max tag, i.e. cost the next split-operand launch a pass over it: two [B, hidden] passes less per block and step."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, lin0, lin1):
        split = _split_mode(x, lin0.in_features)
        ctx.x_max = ctx.t_max = None
        if split:
            ctx.x_max = _lib.tagged_max(x)
            if ctx.x_max is None:
                ctx.x_max = _lib.maxabs(x)
            t = _lib.masked_linear_split(x, lin0._packed_linear_split(), relu_in=True, x_max=ctx.x_max)
            ctx.t_max = _lib.tagged_max(t)
            y = _lib.masked_linear_split(t, lin1._packed_linear_split(), residual=x, relu_in=True, x_max=ctx.t_max)
        else:
            t = _lib.masked_linear(x, lin0._packed_linear(), relu_in=True)
            y = _lib.masked_linear(t, lin1._packed_linear(), residual=x, relu_in=True)
        ctx.save_for_backward(x, t)
        ctx.layers = (lin0, lin1)
        ctx.params = ((w0, b0), (w1, b1))
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, t = ctx.saved_tensors
        lin0, lin1 = ctx.layers
        (w0, b0), (w1, b1) = ctx.params
        grad_y = grad_y.contiguous()
        need = ctx.needs_input_grad
        want0, want1 = need[1] or (b0 is not None and need[2]), need[3] or (b1 is not None and need[4])
        sink1 = _grad_sink(w1, b1, b1 is not None) if want1 else None
        gt, gw1, gb1 = _linear_backward(lin1, t, grad_y, True, ctx.t_max, True, want1, b1 is not None, into=sink1)
        sink0 = _grad_sink(w0, b0, b0 is not None) if want0 else None
        gx, gw0, gb0 = _linear_backward(lin0, x, gt, True, ctx.x_max, need[0], want0, b0 is not None, into=sink0,
                                        add=grad_y)
        if sink1 is not None:
            _sink_done(w1, b1)
            gw1 = gb1 = None
        if sink0 is not None:
            _sink_done(w0, b0)
            gw0 = gb0 = None
        return gx, gw0, gb0, gw1, gb1, None, None


# bf16 training mode: keep the conditioner output of the fused training forward as bf16 (False: fp32, A/B and tests)
KEEP_PARAMS_BF16 = True


class FinalRqsFn(torch.autograd.Function):
    """Final MADE layer + RQ spline of a whole autoregressive layer, forward direction, as ONE launch that also keeps the
    conditioner output (nf_made_rqs_fwd_params): (y, logabsdet) = spline(x; hidden @ (W*mask)^T + b).  Backward: the
    spline's backward kernel on the kept parameters, then the final layer's input / weight gradient GEMMs - what
    MaskedLinearFn + RqsFn record as two nodes, minus one pass over the [B, D*P] parameter tensor."""

    @staticmethod
    def forward(ctx, x, hidden, weight, bias, layer, cfg):
        num_bins, tails, tail_bound, mins = cfg
        pf = layer._packed_fused_final(x.shape[1], num_bins, tails)
        # bf16 mode: the kept parameters have two readers, this node's spline backward and (through it) the two bf16 /
        # fp16 backward GEMMs - they are kept as bf16 (KEEP_PARAMS_BF16): half the bytes stored now and read then
        keep16 = (KEEP_PARAMS_BF16 and _bf16_mode() and _lib.bf16_gradient_path(layer.in_features, layer.out_features)
                  and (x.shape[1] * (3 * num_bins - 1 if tails == 'linear' else 3 * num_bins + 1)) % 8 == 0)
        y, lad, params = _lib.made_rqs_params(hidden, pf, x, tail_bound, min_bin_width=mins[0], min_bin_height=mins[1],
                                              min_derivative=mins[2], keep_bf16=keep16)
        ctx.save_for_backward(x, params, hidden)
        ctx.layer, ctx.cfg, ctx.has_bias = layer, cfg, bias is not None
        ctx.h_max = _lib.tagged_max(hidden)
        ctx.params = (weight, bias)
        return y, lad

    @staticmethod
    def backward(ctx, grad_y, grad_lad):
        x, params, hidden = ctx.saved_tensors
        num_bins, tails, tail_bound, mins = ctx.cfg
        if grad_y is None:
            grad_y = torch.zeros_like(x)
        if grad_lad is None:
            grad_lad = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        layer = ctx.layer
        h_max = ctx.h_max
        if h_max is None and _bf16_mode():
            h_max = _lib.maxabs(hidden)
        want_w = ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3])
        sink = _grad_sink(*ctx.params, ctx.has_bias) if want_w else None
        if _bf16_mode() and _lib.bf16_gradient_path(layer.in_features, layer.out_features):
            # bf16 mode: the parameter gradients never leave this node and both GEMMs below round them to bf16 (the
            # weight gradient: to fp16) anyway - the spline backward writes them as bf16 (half the bytes written and
            # re-read twice), the input gradient takes the plane as its A operand, the weight gradient widens it
            gx, gp16 = _lib.rqs_bwd(x, params, grad_y.contiguous(), grad_lad.contiguous(), num_bins, tails, tail_bound,
                                    min_bin_width=mins[0], min_bin_height=mins[1], min_derivative=mins[2],
                                    bf16_out=True)
            gh = gw = gb = None
            if ctx.needs_input_grad[1]:
                gh = _lib.masked_linear_split_bwd_input_bf16(gp16, layer._packed_linear_split_t())
            if want_w:
                gw, gb = _lib.masked_linear_half_bwd_weight_bf16(hidden, gp16, getattr(layer, 'mask', None),
                                                                 layer._wgrad_ktab(), False, want_bias=ctx.has_bias,
                                                                 x_max=h_max, grad_y_max=_lib.tagged_max(gp16), into=sink)
            if sink is not None:
                _sink_done(*ctx.params)
                gw = gb = None
            return gx, gh, gw, gb, None, None
        gx, gparams = _lib.rqs_bwd(x, params, grad_y.contiguous(), grad_lad.contiguous(), num_bins, tails, tail_bound,
                                   min_bin_width=mins[0], min_bin_height=mins[1], min_derivative=mins[2],
                                   want_max=_f16x2_mode() or _bf16_mode())
        gh, gw, gb = _linear_backward(ctx.layer, hidden, gparams, False, h_max, ctx.needs_input_grad[1], want_w,
                                      ctx.has_bias, into=sink)
        if sink is not None:
            _sink_done(*ctx.params)
            gw = gb = None
        return gx, gh, gw, gb, None, None


class StdNormalLogProbFn(torch.autograd.Function):
    """log N(z; 0, I) + sign * logabsdet per row as one launch under autograd as well (nf_std_normal_log_prob; the
    reference's pow / mul / add / sum / add chain and its five backward nodes otherwise: distributions.py:27-36 there).
    Backward: d/dz = -z * g per row, d/dlogabsdet = sign * g."""

    @staticmethod
    def forward(ctx, z, logabsdet, sign):
        ctx.save_for_backward(z)
        ctx.sign, ctx.has_lad = float(sign), logabsdet is not None
        return _lib.std_normal_log_prob(z, logabsdet, sign)

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        gz = z * (-g).unsqueeze(1) if ctx.needs_input_grad[0] else None
        gl = g * ctx.sign if ctx.has_lad and ctx.needs_input_grad[1] else None
        return gz, gl, None


class RqsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, num_bins, tails, tail_bound, mins, pre_div=1.0, inverse=False):
        y, lad, _ = _lib.rqs(x, params, num_bins, tails, tail_bound, inverse=inverse, pre_div=pre_div,
                             min_bin_width=mins[0], min_bin_height=mins[1], min_derivative=mins[2])
        # the inverse direction is differentiated implicitly at its OUTPUT (nf_rqs_inverse_bwd): that is what it keeps
        ctx.save_for_backward(y if inverse else x, params)
        ctx.cfg = (num_bins, tails, tail_bound, mins, pre_div, inverse)
        return y, lad

    @staticmethod
    def backward(ctx, grad_y, grad_lad):
        x, params = ctx.saved_tensors
        num_bins, tails, tail_bound, mins, pre_div, inverse = ctx.cfg
        if grad_y is None:
            grad_y = torch.zeros_like(x)
        if grad_lad is None:
            grad_lad = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        gx, gparams = _lib.rqs_bwd(x, params, grad_y.contiguous(), grad_lad.contiguous(), num_bins, tails, tail_bound,
                                   pre_div=pre_div, min_bin_width=mins[0], min_bin_height=mins[1],
                                   min_derivative=mins[2], inverse=inverse, want_max=_f16x2_mode() or _bf16_mode())
        return gx, gparams, None, None, None, None, None, None


class FamilySplineFn(torch.autograd.Function):
    """The linear / quadratic / cubic spline (nf_*_spline_fwd) with its backward kernel (nf_*_spline_bwd), either
    direction; cfg = dict(num_bins, tails, tail_bound, pre_div, min_bin_width, min_bin_height)."""

    @staticmethod
    def forward(ctx, x, params, kind, cfg, inverse=False):
        y, lad, _ = _lib.family_spline(kind, x, params, inverse=inverse, **cfg)
        ctx.save_for_backward(y if inverse else x, params)       # the inverse is differentiated at its OUTPUT
        ctx.cfg = (kind, cfg, inverse)
        return y, lad

    @staticmethod
    def backward(ctx, grad_y, grad_lad):
        x, params = ctx.saved_tensors
        kind, cfg, inverse = ctx.cfg
        if grad_y is None:
            grad_y = torch.zeros_like(x)
        if grad_lad is None:
            grad_lad = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        gx, gparams = _lib.family_spline_bwd(kind, x, params, grad_y.contiguous(), grad_lad.contiguous(),
                                             inverse=inverse, **cfg)
        return gx, gparams, None, None, None


class AffineFn(torch.autograd.Function):
    """nf_affine_fwd / nf_affine_bwd.  `feat_index` (int32, coupling) selects the transformed columns; the other
    columns of the output are the input's, and their gradient passes through."""

    @staticmethod
    def forward(ctx, x, params, layout, inverse, feat_index, num_features):
        out = x.clone() if feat_index is not None else torch.empty_like(x)
        y, lad = _lib.affine(x, params, layout, inverse=inverse, feat_index=feat_index, out=out,
                             num_features=num_features)
        ctx.save_for_backward(x, params)
        ctx.cfg = (layout, inverse, feat_index, num_features)
        return y, lad

    @staticmethod
    def backward(ctx, grad_y, grad_lad):
        x, params = ctx.saved_tensors
        layout, inverse, feat_index, num_features = ctx.cfg
        if grad_y is None:
            grad_y = torch.zeros_like(x)
        gx, gparams = _lib.affine_bwd(x, params, layout, grad_y.contiguous(), grad_lad, inverse=inverse,
                                      feat_index=feat_index, num_features=num_features)
        return gx, gparams, None, None, None, None
