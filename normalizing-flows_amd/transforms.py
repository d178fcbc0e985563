"""Flow transforms of the hot path, API-compatible with the reference's models/transforms.py.

    forward(inputs, context=None) -> (outputs, logabsdet[B])
    inverse(inputs, context=None) -> (outputs, logabsdet[B])

Each elementwise transform (+ its per-sample log|det J| reduction) is ONE HIP kernel
(nf_rqs_fwd / nf_affine_fwd); the conditioners are the fused MFMA layers of ``made.py`` /
``resnet.py``.  Submodule and buffer names match the reference so its ``state_dict`` loads as is.
"""
import warnings

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import _lib
from . import made as made_module
from .resnet import ResidualNet

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3

# Rows evaluated per launch group.  The conditioner output of one D=784, K=8 layer is 72 KB per
# sample; 131072 rows keep that intermediate at 9.5 GB (of 288 GB) while every launch still fills the
# 256 CUs many times over (few, large launches: small grids lose a third of the chip to tail effects).
CHUNK_ROWS = 131072


# False: always take the windowed incremental inverse (A/B measurements, tests of that path)
ROW_RESIDENT_INVERSE = True
# False: the f16x2 mode runs the hidden stack as one launch per layer (A/B measurements, tests of that path)
FUSED_HIDDEN_STACK = True
# False: RQ coupling layers run as hidden-stack launch + fused final layer launch (A/B measurements, tests of that path)
FUSED_COUPLING_LAYER = True
# training forward of an autoregressive RQ layer as one launch that keeps the conditioner output (False: GEMM + spline
# as two autograd nodes; A/B and tests)
FUSED_TRAINING_FORWARD = True
# ... in the f16x2 mode too (the t4 kernel storing its parameter tile).  Off: measured on one box, B = 32768, the launch
# takes 1.995 ms against 1.50 + 0.56 ms for GEMM + spline kernel, but the plane split, the hidden-state max pass and the
# pack's table scan around it make the step 0.8 ms (2 %) SLOWER than the two-node graph.
FUSED_TRAINING_FORWARD_F16X2 = False


def _check_2d(inputs, features):
    if inputs.dim() != 2:
        raise ValueError('Inputs must be a 2D tensor.')   # image (4D) inputs are outside this path
    if inputs.shape[1] != features:
        raise ValueError('Expected features = {}, got {}.'.format(features, inputs.shape[1]))


class CompositeTransform(nn.Module):
    """Applies transforms in order, summing logabsdet (reference transforms.py:14-36).  The
    accumulator is allocated like the input (device and dtype), not in the default dtype."""

    def __init__(self, transforms):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)

    @staticmethod
    def _cascade(inputs, funcs, context):
        outputs = inputs
        total_logabsdet = torch.zeros(inputs.shape[0], dtype=inputs.dtype, device=inputs.device)
        for func in funcs:
            outputs, logabsdet = func(outputs, context)
            total_logabsdet += logabsdet
        return outputs, total_logabsdet

    def forward(self, inputs, context=None):
        return self._cascade(inputs, self._transforms, context)

    def inverse(self, inputs, context=None):
        funcs = (transform.inverse for transform in self._transforms[::-1])
        return self._cascade(inputs, funcs, context)


# ------------------------------------------------------------------------------------ autoregressive
class AutoregressiveTransform(nn.Module):
    """Elementwise transform whose parameters come from a MADE over the earlier features
    (reference transforms.py:475-509).  forward = one conditioner pass; inverse = D passes."""

    def __init__(self, autoregressive_net):
        super().__init__()
        self.autoregressive_net = autoregressive_net

    def _recording(self, inputs):
        return torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward(self, inputs, context=None):
        _check_2d(inputs, self.autoregressive_net.initial_layer.in_features)
        if self._recording(inputs):
            return self._forward_autograd(inputs, context)
        B = inputs.shape[0]
        outputs = torch.empty_like(inputs)
        logabsdet = torch.empty(B, dtype=inputs.dtype, device=inputs.device)
        if B <= CHUNK_ROWS or any(isinstance(m, nn.BatchNorm1d) and m.training for m in self.modules()):
            # one launch group (always with BatchNorm in train() mode: its statistics are the whole batch's): no slicing, so the tensors keep what the kernels attached to them (the
            # split-operand kernels hand max |y| to the next layer's first GEMM that way)
            self._transform_rows(inputs, inputs, context, outputs, logabsdet, inverse=False)
            return outputs, logabsdet
        # per launch group; what the kernels attach to a group's output (max |y| of the split-operand kernels)
        # is kept per group on the full tensor and re-attached to the next layer's slices
        # (valid only for the storage and version they were taken at: an in-place write in between drops them)
        have = getattr(inputs, '_nf_chunk_max', None)
        have = have[0] if have is not None and have[1:] == (inputs._version, inputs.data_ptr()) else {}
        keep = {}
        for s in range(0, B, CHUNK_ROWS):
            e = min(B, s + CHUNK_ROWS)
            ctx = None if context is None else context[s:e]
            xin, yout = inputs[s:e], outputs[s:e]
            _lib.tag_max(xin, have.get((s, e)))
            self._transform_rows(xin, xin, ctx, yout, logabsdet[s:e], inverse=False)
            keep[(s, e)] = _lib.tagged_max(yout)
        outputs._nf_chunk_max = (keep, outputs._version, outputs.data_ptr())
        return outputs, logabsdet

    def _forward_autograd(self, inputs, context):
        """Training path: conditioner through MaskedLinearFn, the elementwise transform through the subclass's
        autograd.Function (the parameter tensor is materialised: its backward needs it)."""
        params = self.autoregressive_net(inputs.contiguous(), context)
        return self._elementwise_autograd(inputs.contiguous(), params, inverse=False)

    def _inverse_autograd(self, inputs, context):
        """The sampling direction under autograd (what the reference's default reverse-KL loss differentiates,
        model.py:135-139 through :78-85): the reference's own D conditioner passes (transforms.py:493-500), every one of
        them recorded - conditioner GEMMs as MaskedLinearFn, the elementwise inverse as this transform's
        `_elementwise_autograd`.  Memory grows with D like the reference's; meant for the low-dimensional flows the
        reference trains this way (its scripts use D = 2)."""
        D = inputs.shape[1]
        inputs = inputs.contiguous()
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        for _ in range(D):
            params = self.autoregressive_net(outputs, context)
            outputs, logabsdet = self._elementwise_autograd(inputs, params, inverse=True)
        return outputs, logabsdet

    def _elementwise_autograd(self, inputs, params, inverse):
        raise NotImplementedError('this transform has no backward kernels: call under torch.no_grad()')

    def inverse(self, inputs, context=None):
        _check_2d(inputs, self.autoregressive_net.initial_layer.in_features)
        if self._recording(inputs):
            return self._inverse_autograd(inputs, context)
        B, D = inputs.shape
        outputs = torch.zeros_like(inputs)
        logabsdet = torch.zeros(B, dtype=inputs.dtype, device=inputs.device)
        for s in range(0, max(B, 1), CHUNK_ROWS):
            e = min(B, s + CHUNK_ROWS)
            ctx = None if context is None else context[s:e]
            cur = outputs[s:e]
            nxt = torch.empty_like(cur)
            for _ in range(D):
                self._transform_rows(cur, inputs[s:e], ctx, nxt, logabsdet[s:e], inverse=True)
                cur, nxt = nxt, cur
            if cur.data_ptr() != outputs[s:e].data_ptr():
                outputs[s:e].copy_(cur)
        return outputs, logabsdet

    def _inverse_via_rows(self, inputs, context, kind, num_bins, tails, rows_kernel):
        """Same result as the D full passes above.  With hidden units in degree order one launch carries 32 samples
        per workgroup through features 0..H (nf_inverse.hip, see the RQ-spline transform below); these elementwise
        inverses depend on earlier features only through the parameters, so the final hidden state that launch leaves
        behind gives every feature and logabsdet in one final-layer GEMM + one elementwise pass."""
        net = self.autoregressive_net
        if self._recording(inputs):
            _check_2d(inputs, net.initial_layer.in_features)
            return self._inverse_autograd(inputs, context)
        pi = net.inverse_rows_pack(num_bins, tails, kind=kind) if ROW_RESIDENT_INVERSE else None
        if pi is None:
            return AutoregressiveTransform.inverse(self, inputs, context)
        D = net.initial_layer.in_features
        _check_2d(inputs, D)
        made_module._require_inference(net.final_layer)
        B, H = inputs.shape[0], net.final_layer.in_features
        outputs = torch.zeros_like(inputs)
        logabsdet = torch.zeros(B, dtype=inputs.dtype, device=inputs.device)
        for s in range(0, B, CHUNK_ROWS):
            e = min(B, s + CHUNK_ROWS)
            hidden = torch.empty((e - s, H), dtype=inputs.dtype, device=inputs.device)
            ctx = net.inverse_rows_context(None if context is None else context[s:e])
            rows_kernel(pi, inputs[s:e], outputs[s:e], hidden, ctx)
            self._elementwise(inputs[s:e], net.final_layer.fused(hidden), outputs[s:e], logabsdet[s:e], inverse=True)
        return outputs, logabsdet

    def _transform_rows(self, net_inputs, inputs, context, outputs, logabsdet, inverse):
        """Conditioner on `net_inputs`, elementwise transform of `inputs` into outputs / logabsdet."""
        params = self.autoregressive_net(net_inputs, context)
        self._elementwise(inputs, params, outputs, logabsdet, inverse)

    def _output_dim_multiplier(self):
        raise NotImplementedError()

    def _elementwise(self, inputs, params, outputs, logabsdet, inverse):
        raise NotImplementedError()


class MaskedAffineAutoregressiveTransform(AutoregressiveTransform):
    """scale = sigmoid(u+2)+1e-3, y = scale*x + shift (reference transforms.py:511-566)."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2, use_residual_blocks=True,
                 random_mask=False, activation=F.relu, dropout_probability=0., use_batch_norm=False):
        self.features = features
        net = made_module.MADE(features=features, hidden_features=hidden_features,
                               context_features=context_features, num_blocks=num_blocks,
                               output_multiplier=self._output_dim_multiplier(),
                               use_residual_blocks=use_residual_blocks, random_mask=random_mask,
                               activation=activation, dropout_probability=dropout_probability,
                               use_batch_norm=use_batch_norm)
        super().__init__(net)

    def _output_dim_multiplier(self):
        return 2

    def _elementwise(self, inputs, params, outputs, logabsdet, inverse):
        _lib.affine(inputs, params, _lib.NF_AFFINE_INTERLEAVED, inverse=inverse, out=outputs, logabsdet=logabsdet)

    def _forward_autograd(self, inputs, context):
        """Training path: conditioner (MaskedLinearFn) and the affine map (AffineFn) over the backward kernels."""
        x = inputs.contiguous()
        return self._elementwise_autograd(x, self.autoregressive_net(x, context), inverse=False)

    def _elementwise_autograd(self, inputs, params, inverse):
        from .autograd import AffineFn
        return AffineFn.apply(inputs, params.contiguous(), _lib.NF_AFFINE_INTERLEAVED, bool(inverse), None, None)

    def inverse(self, inputs, context=None):
        return self._inverse_via_rows(inputs, context, 'affine', None, None,
                                      lambda pi, z, out, hid, ctx: _lib.made_affine_inverse_rows(pi, z, out, hid, ctx=ctx))


class MaskedPiecewiseRationalQuadraticAutoregressiveTransform(AutoregressiveTransform):
    """Rational-quadratic spline MAF layer (reference transforms.py:771-866).  As in the reference the
    MADE conditioner has no `hidden_features` attribute, so widths/heights are NOT rescaled."""

    def __init__(self, features, hidden_features, context_features=None, num_bins=10, tails=None, tail_bound=1.,
                 num_blocks=2, use_residual_blocks=True, random_mask=False, activation=F.relu,
                 dropout_probability=0., use_batch_norm=False, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=DEFAULT_MIN_BIN_HEIGHT, min_derivative=DEFAULT_MIN_DERIVATIVE):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound
        net = made_module.MADE(features=features, hidden_features=hidden_features,
                               context_features=context_features, num_blocks=num_blocks,
                               output_multiplier=self._output_dim_multiplier(),
                               use_residual_blocks=use_residual_blocks, random_mask=random_mask,
                               activation=activation, dropout_probability=dropout_probability,
                               use_batch_norm=use_batch_norm)
        super().__init__(net)

    def _output_dim_multiplier(self):
        if self.tails == 'linear':
            return self.num_bins * 3 - 1
        elif self.tails is None:
            return self.num_bins * 3 + 1
        else:
            raise ValueError

    def _elementwise(self, inputs, params, outputs, logabsdet, inverse, want_bins=False):
        return _lib.rqs(inputs, params, self.num_bins, self.tails, self.tail_bound, inverse=inverse, out=outputs,
                        logabsdet=logabsdet, min_bin_width=self.min_bin_width,
                        min_bin_height=self.min_bin_height, min_derivative=self.min_derivative,
                        want_bins=want_bins)

    def _fusable(self):
        net = self.autoregressive_net
        return _lib.made_rqs_supported(self.num_bins, self.tails, net.final_layer.in_features)

    def _transform_rows(self, net_inputs, inputs, context, outputs, logabsdet, inverse, want_bins=False):
        """One launch for the final MADE layer and the spline (nf_made_rqs_fwd): the [B, D*P] parameter
        tensor of the reference is never written.  Spline shapes the fused kernel does not cover
        (K < 4, hidden width not a multiple of 8) take the two-kernel route."""
        net = self.autoregressive_net
        if not self._fusable():
            params = net(net_inputs, context)
            return self._elementwise(inputs, params, outputs, logabsdet, inverse, want_bins=want_bins)
        made_module._require_inference(net.final_layer)
        pf = net.final_layer._packed_fused_final(net.initial_layer.in_features, self.num_bins, self.tails)
        if FUSED_HIDDEN_STACK and made_module.get_matmul_dtype() == 'f16x2' and context is None:
            # two launches per layer: the whole hidden stack (nf_made_hidden_planes_fwd: the hidden state leaves the CU
            # only as the fp16 term planes the final layer reads, one scale per 64 rows), then final layer + spline
            ph = net.hidden_planes_pack()
            if ph is not None:
                planes, row_scale, _ = _lib.made_hidden_planes(net_inputs, ph)
                return _lib.made_rqs_planes(planes, row_scale, pf, inputs, self.tail_bound, inverse=inverse,
                                            out=outputs, logabsdet=logabsdet, min_bin_width=self.min_bin_width,
                                            min_bin_height=self.min_bin_height, min_derivative=self.min_derivative,
                                            want_bins=want_bins)
        hidden = net.forward_hidden(net_inputs, context)
        return _lib.made_rqs(hidden, pf, inputs, self.tail_bound, inverse=inverse, out=outputs, logabsdet=logabsdet,
                             min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                             min_derivative=self.min_derivative, want_bins=want_bins)

    def _forward_autograd(self, inputs, context):
        """Training path: conditioner and spline as autograd.Functions over the backward kernels.  The
        parameter tensor is materialised here (the backward needs it); inference uses the fused kernel."""
        net = self.autoregressive_net
        inputs = inputs.contiguous()
        mode = made_module.get_matmul_dtype()
        if self._fusable() and FUSED_TRAINING_FORWARD and \
                (mode in ('f32', 'bf16') or (mode == 'f16x2' and FUSED_TRAINING_FORWARD_F16X2)) and \
                _lib.made_rqs_params_supported(self.num_bins, self.tails, made_module._MATMUL_DTYPE[mode]):
            # one launch for the final layer and the spline that also writes the parameters its backward recomputes
            # from (nf_made_rqs_fwd_params): they are not read back between the GEMM and the spline
            from .autograd import FinalRqsFn
            final = net.final_layer
            hidden = net.forward_hidden(inputs, context)
            return FinalRqsFn.apply(inputs, hidden.contiguous(), final.weight, final.bias, final,
                                    (self.num_bins, self.tails, self.tail_bound,
                                     (self.min_bin_width, self.min_bin_height, self.min_derivative)))
        params = net(inputs, context)      # context: extra GEMM + GLU gates (torch glue)
        return self._elementwise_autograd(inputs, params, inverse=False)

    def _elementwise_autograd(self, inputs, params, inverse):
        from .autograd import RqsFn
        return RqsFn.apply(inputs, params.contiguous(), self.num_bins, self.tails, self.tail_bound,
                           (self.min_bin_width, self.min_bin_height, self.min_derivative), 1.0, bool(inverse))

    def _feature_groups(self):
        """Consecutive features that can be inverted in the same pass.  Feature i (0-based, MADE degree
        i+1) sees hidden units of degree <= i, and input i-1 reaches it only through units of degree
        exactly i; where no hidden unit has that degree, feature i does not depend on feature i-1 and
        joins its pass.  D=784, H=512: 513 single-feature passes, then features 513..783 together."""
        net = self.autoregressive_net
        key = tuple(net.hidden_degrees())
        c = getattr(self, '_groups_cache', None)
        if c is None or c[0] != key:
            D = net.initial_layer.in_features
            present = set(key)
            groups, a = [], 0
            for i in range(1, D):
                if i in present:
                    groups.append((a, i - 1))
                    a = i
            groups.append((a, D - 1))
            object.__setattr__(self, '_groups_cache', (key, groups))
        return self._groups_cache[1]

    def inverse(self, inputs, context=None):
        """Same result as the reference's D full passes (transforms.py:493-500), computed incrementally:
        after the pass that fixes feature i-1 only the hidden units of degree i can change among those
        later features still read, so a pass recomputes just their 128-column tiles in every hidden
        layer and the one fused final-layer tile that holds the features being fixed.  Features not yet
        fixed stay exactly zero in the network input, as in the reference's first pass."""
        net = self.autoregressive_net
        if self._recording(inputs):
            _check_2d(inputs, net.initial_layer.in_features)
            return self._inverse_autograd(inputs, context)
        if not net.relu_only() or not self._fusable() or (context is not None and not (ROW_RESIDENT_INVERSE and net.inverse_rows_pack(self.num_bins, self.tails))):
            return super().inverse(inputs, context)
        D = net.initial_layer.in_features
        _check_2d(inputs, D)
        made_module._require_inference(net.final_layer)
        B = inputs.shape[0]
        H = net.final_layer.in_features
        pf = net.final_layer._packed_fused_final(D, self.num_bins, self.tails)
        fpt = 256 // self._output_dim_multiplier()        # whole features per fused column tile
        outputs = torch.zeros_like(inputs)
        logabsdet = torch.zeros(B, dtype=inputs.dtype, device=inputs.device)
        pi = net.inverse_rows_pack(self.num_bins, self.tails) if ROW_RESIDENT_INVERSE else None
        if pi is not None:
            # hidden units in degree order: 32 samples per workgroup run features 0..H start to end in one launch
            # (nf_inverse.hip, about one forward pass of work); the features past H+1 depend on no further hidden
            # unit and take one windowed launch of the fused layer kernel on the final hidden state
            t0 = (H + 1) // fpt
            for s in range(0, B, CHUNK_ROWS):
                e = min(B, s + CHUNK_ROWS)
                z, out = inputs[s:e], outputs[s:e]
                hidden = torch.empty((e - s, H), dtype=inputs.dtype, device=inputs.device)
                ws = _lib.fused_workspace(pf, e - s, inputs.device, planes=False)
                ctx = net.inverse_rows_context(None if context is None else context[s:e])
                _lib.made_rqs_inverse_rows(pi, z, out, hidden, ws, self.tail_bound, self.min_bin_width,
                                           self.min_bin_height, self.min_derivative, ctx=ctx)
                _lib.made_rqs(hidden, pf, z, self.tail_bound, inverse=True, out=out, logabsdet=logabsdet[s:e],
                              min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                              min_derivative=self.min_derivative, tiles=(t0, pf.tiles - t0), feat_limit=D,
                              reduce=True, workspace=ws)
            return outputs, logabsdet
        groups = self._feature_groups()
        names = {n for st in net.hidden_program()[0] for n in (st[1], st[2])} - {'x'}
        for s in range(0, B, CHUNK_ROWS):
            e = min(B, s + CHUNK_ROWS)
            z, out = inputs[s:e], outputs[s:e]
            bufs = {n: torch.empty((e - s, H), dtype=inputs.dtype, device=inputs.device) for n in names}
            bufs['x'] = out
            ws = _lib.fused_workspace(pf, e - s, inputs.device, planes=False)
            for gi, (a, b) in enumerate(groups):
                hidden = net.run_hidden_program(bufs, degree=None if gi == 0 else a)
                t0, t1 = a // fpt, b // fpt
                _lib.made_rqs(hidden, pf, z, self.tail_bound, inverse=True, out=out, logabsdet=logabsdet[s:e],
                              min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                              min_derivative=self.min_derivative, tiles=(t0, t1 - t0 + 1), feat_limit=b + 1,
                              reduce=gi == len(groups) - 1, workspace=ws)
        return outputs, logabsdet

    def forward_with_bins(self, inputs, context=None):
        """forward() that also returns the int32 bin index of every element (-1 in the tails): the
        searchsorted result (helper_functions.py:507-512) that parity tests compare bit for bit."""
        return self._transform_rows(inputs, inputs, context, torch.empty_like(inputs), None, False, want_bins=True)


class MaskedPiecewiseLinearAutoregressiveTransform(AutoregressiveTransform):
    """Piecewise-linear CDF spline on [0, 1] with MADE-produced bin masses (reference transforms.py:568-616;
    its `splines.linear_spline` is read as models/splines/linear.py:35-94)."""

    def __init__(self, num_bins, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu, dropout_probability=0.,
                 use_batch_norm=False):
        self.num_bins = num_bins
        self.features = features
        made = made_module.MADE(
            features=features, hidden_features=hidden_features, context_features=context_features,
            num_blocks=num_blocks, output_multiplier=self._output_dim_multiplier(),
            use_residual_blocks=use_residual_blocks, random_mask=random_mask, activation=activation,
            dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        return self.num_bins

    def _elementwise(self, inputs, params, outputs, logabsdet, inverse):
        _lib.linear_spline(inputs, params, self.num_bins, tails=None, inverse=inverse, out=outputs,
                           logabsdet=logabsdet)

    def _elementwise_autograd(self, inputs, params, inverse):
        from .autograd import FamilySplineFn
        return FamilySplineFn.apply(inputs, params.contiguous(), 'linear', dict(num_bins=self.num_bins), bool(inverse))

    def inverse(self, inputs, context=None):
        return self._inverse_via_rows(inputs, context, 'linear', self.num_bins, None,
                                      lambda pi, z, out, hid, ctx: _lib.made_family_inverse_rows(
                                          pi, 'linear', z, out, hid, ctx=ctx))


class MaskedPiecewiseQuadraticAutoregressiveTransform(AutoregressiveTransform):
    """Piecewise-quadratic CDF spline with MADE-produced widths / knot heights (reference transforms.py:618-705;
    `tails=None` -> models/splines/quadratic.py:48-135 on [0, 1], 'linear' -> :11-46).  The reference's
    division by sqrt(hidden_features) never fires on this path: MADE has no such attribute."""

    def __init__(self, features, hidden_features, context_features=None, num_bins=10, num_blocks=2, tails=None,
                 tail_bound=1., use_residual_blocks=True, random_mask=False, activation=F.relu,
                 dropout_probability=0., use_batch_norm=False, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=DEFAULT_MIN_BIN_HEIGHT, min_derivative=DEFAULT_MIN_DERIVATIVE):
        if tails not in (None, 'linear'):
            raise ValueError
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound
        self.features = features
        made = made_module.MADE(
            features=features, hidden_features=hidden_features, context_features=context_features,
            num_blocks=num_blocks, output_multiplier=self._output_dim_multiplier(),
            use_residual_blocks=use_residual_blocks, random_mask=random_mask, activation=activation,
            dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        return self.num_bins * 2 - 1 if self.tails == 'linear' else self.num_bins * 2 + 1

    def _elementwise(self, inputs, params, outputs, logabsdet, inverse):
        if self.min_bin_width * self.num_bins > 1.0:
            raise ValueError('Minimal bin width too large for the number of bins')
        if self.min_bin_height * self.num_bins > 1.0:
            raise ValueError('Minimal bin height too large for the number of bins')
        _lib.quadratic_spline(inputs, params, self.num_bins, tails=self.tails, tail_bound=self.tail_bound,
                              inverse=inverse, out=outputs, logabsdet=logabsdet, min_bin_width=self.min_bin_width,
                              min_bin_height=self.min_bin_height)

    def _elementwise_autograd(self, inputs, params, inverse):
        from .autograd import FamilySplineFn
        cfg = dict(num_bins=self.num_bins, tails=self.tails, tail_bound=self.tail_bound,
                   min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height)
        return FamilySplineFn.apply(inputs, params.contiguous(), 'quadratic', cfg, bool(inverse))

    def inverse(self, inputs, context=None):
        return self._inverse_via_rows(inputs, context, 'quadratic', self.num_bins, self.tails,
                                      lambda pi, z, out, hid, ctx: _lib.made_family_inverse_rows(
                                          pi, 'quadratic', z, out, hid, tail_bound=self.tail_bound,
                                          min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height, ctx=ctx))


class MaskedPiecewiseCubicAutoregressiveTransform(AutoregressiveTransform):
    """Monotone piecewise-cubic spline on [-10, 10] (the default interval of models/splines/cubic.py:56-67, which
    is what reference transforms.py:707-769 calls) with MADE-produced widths, heights and boundary derivatives."""

    def __init__(self, num_bins, features, hidden_features, context_features=None, num_blocks=2,
                 use_residual_blocks=True, random_mask=False, activation=F.relu, dropout_probability=0.,
                 use_batch_norm=False):
        self.num_bins = num_bins
        self.features = features
        made = made_module.MADE(
            features=features, hidden_features=hidden_features, context_features=context_features,
            num_blocks=num_blocks, output_multiplier=self._output_dim_multiplier(),
            use_residual_blocks=use_residual_blocks, random_mask=random_mask, activation=activation,
            dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        return self.num_bins * 2 + 2

    def _elementwise(self, inputs, params, outputs, logabsdet, inverse):
        _lib.cubic_spline(inputs, params, self.num_bins, tails=None, bound=10.0, inverse=inverse, out=outputs,
                          logabsdet=logabsdet)

    def _elementwise_autograd(self, inputs, params, inverse):
        from .autograd import FamilySplineFn
        return FamilySplineFn.apply(inputs, params.contiguous(), 'cubic', dict(num_bins=self.num_bins, tail_bound=10.0),
                                    bool(inverse))

    def inverse(self, inputs, context=None):
        return self._inverse_via_rows(inputs, context, 'cubic', self.num_bins, None,
                                      lambda pi, z, out, hid, ctx: _lib.made_family_inverse_rows(
                                          pi, 'cubic', z, out, hid, tail_bound=10.0, ctx=ctx))


# ------------------------------------------------------------------------------------ coupling
class CouplingTransform(nn.Module):
    """Splits features by a binary mask; the identity half conditions the transform of the other
    half (reference transforms.py:38-147).  2-D inputs only on this path."""

    def __init__(self, mask, transform_net_create_fn, unconditional_transform=None):
        mask = torch.as_tensor(mask)
        if mask.dim() != 1:
            raise ValueError('Mask must be a 1-dim tensor.')
        if mask.numel() <= 0:
            raise ValueError('Mask can\'t be empty.')
        super().__init__()
        self.features = len(mask)
        idx = torch.arange(self.features)
        self.register_buffer('identity_features', idx.masked_select(mask <= 0))
        self.register_buffer('transform_features', idx.masked_select(mask > 0))
        assert self.num_identity_features + self.num_transform_features == self.features
        self.transform_net = transform_net_create_fn(
            self.num_identity_features, self.num_transform_features * self._transform_dim_multiplier())
        # optional transform of the identity half itself (reference transforms.py:65-70, :98-101, :120-123)
        self.unconditional_transform = None if unconditional_transform is None else \
            unconditional_transform(features=self.num_identity_features)

    @property
    def num_identity_features(self):
        return len(self.identity_features)

    @property
    def num_transform_features(self):
        return len(self.transform_features)

    def _index32(self, name):
        # int32 copy of an index buffer, for the kernels
        t = getattr(self, name)
        key = (t.device, t._version, t.data_ptr())
        cache = self.__dict__.setdefault('_index_cache', {})
        c = cache.get(name)
        if c is None or c[0] != key:
            c = cache[name] = (key, t.to(torch.int32).contiguous())
        return c[1]

    def _feat_index(self):
        return self._index32('transform_features')

    def _recording(self, inputs):
        return torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))

    def _run_autograd(self, inputs, context, inverse):
        raise NotImplementedError('this coupling transform has no backward kernels: call under torch.no_grad()')

    def _run(self, inputs, context, inverse):
        _check_2d(inputs, self.features)
        if not inputs.is_contiguous():
            inputs = inputs.contiguous()
        if self._recording(inputs):
            return self._run_autograd(inputs, context, inverse)
        identity_split = inputs[:, self.identity_features].contiguous()
        transform_params = self.transform_net(identity_split, context)
        # identity columns pass through; the kernel overwrites the transformed columns in place order
        outputs = inputs.clone()
        logabsdet = torch.empty(inputs.shape[0], dtype=inputs.dtype, device=inputs.device)
        self._coupling_elementwise(inputs, transform_params, outputs, logabsdet, inverse)
        return outputs, logabsdet

    def forward(self, inputs, context=None):
        if self.unconditional_transform is None:
            return self._run(inputs, context, inverse=False)
        # the conditioner reads the ORIGINAL identity half; its own transform is applied to the output afterwards
        outputs, logabsdet = self._run(inputs, context, inverse=False)
        identity_split, logabsdet_identity = self.unconditional_transform(
            inputs[:, self.identity_features].contiguous(), context)
        return outputs.index_copy(1, self.identity_features, identity_split), logabsdet + logabsdet_identity

    def inverse(self, inputs, context=None):
        if self.unconditional_transform is None:
            return self._run(inputs, context, inverse=True)
        # undo the identity half's own transform first: the conditioner must see what it saw in forward()
        _check_2d(inputs, self.features)
        identity_split, logabsdet_identity = self.unconditional_transform.inverse(
            inputs[:, self.identity_features].contiguous(), context)
        outputs, logabsdet = self._run(inputs.index_copy(1, self.identity_features, identity_split), context,
                                       inverse=True)
        return outputs, logabsdet_identity + logabsdet

    def _transform_dim_multiplier(self):
        raise NotImplementedError()

    def _coupling_elementwise(self, inputs, transform_params, outputs, logabsdet, inverse):
        raise NotImplementedError()


class AffineCouplingTransform(CouplingTransform):
    """shift = p[:, :d], scale = sigmoid(p[:, d:] + 2) + 1e-3 (reference transforms.py:323-347)."""

    def _transform_dim_multiplier(self):
        return 2

    def _coupling_elementwise(self, inputs, transform_params, outputs, logabsdet, inverse):
        _lib.affine(inputs, transform_params, _lib.NF_AFFINE_BLOCKED, inverse=inverse, feat_index=self._feat_index(),
                    out=outputs, logabsdet=logabsdet, num_features=self.num_transform_features)

    def _run_autograd(self, inputs, context, inverse):
        """Training path (either direction): the conditioner through its own autograd route, the affine map and its
        backward as AffineFn (nf_affine_bwd); identity columns pass values and gradients through."""
        from .autograd import AffineFn
        identity_split = inputs[:, self.identity_features].contiguous()
        transform_params = self.transform_net(identity_split, context)
        return AffineFn.apply(inputs, transform_params.contiguous(), _lib.NF_AFFINE_BLOCKED, bool(inverse),
                              self._feat_index(), self.num_transform_features)


class PiecewiseRationalQuadraticCouplingTransform(CouplingTransform):
    """RQ-spline coupling layer (reference transforms.py:173-247).  When the conditioner exposes
    `hidden_features` (ResidualNet does) widths and heights are divided by its square root."""

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.,
                 apply_unconditional_transform=False, img_shape=None, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=DEFAULT_MIN_BIN_HEIGHT, min_derivative=DEFAULT_MIN_DERIVATIVE):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound
        if img_shape:
            raise NotImplementedError('image (4-D) inputs are outside this path')
        unconditional_transform = None
        if apply_unconditional_transform:
            unconditional_transform = lambda features: PiecewiseRationalQuadraticCDF(       # noqa: E731
                shape=[features], num_bins=num_bins, tails=tails, tail_bound=tail_bound, min_bin_width=min_bin_width,
                min_bin_height=min_bin_height, min_derivative=min_derivative)
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        if self.tails == 'linear':
            return self.num_bins * 3 - 1
        else:
            return self.num_bins * 3 + 1

    def _pre_div(self):
        if hasattr(self.transform_net, 'hidden_features'):
            return np.sqrt(self.transform_net.hidden_features)
        if hasattr(self.transform_net, 'hidden_channels'):
            return np.sqrt(self.transform_net.hidden_channels)
        warnings.warn('Inputs to the softmax are not scaled down: initialization might be bad.')
        return 1.0

    def _packed_coupling_final(self):
        """The conditioner's final Linear for the one-launch layer kernel (nf_coupling_final_pack), cached."""
        fl = self.transform_net.final_layer
        key = (_lib.weight_version(fl.weight), fl.weight.data_ptr(), None if fl.bias is None else _lib.weight_version(fl.bias),
               fl.weight.device,
               self.num_bins, self.tails, float(self._pre_div()))
        cached = self.__dict__.get('_coupling_final_cache')
        if cached is None or cached[0] != key:
            cached = (key, _lib.PackedCouplingFinal(fl.weight, fl.bias, self.num_transform_features, self.num_bins,
                                                    self.tails, self._pre_div()))
            self.__dict__['_coupling_final_cache'] = cached
        return cached[1]

    def _fusable(self, inputs):
        """The conditioner's final Linear + the spline run as one launch (nf_coupling_rqs_fwd) when the
        conditioner is this package's ResidualNet in inference mode and the spline shape fits the kernel."""
        net = self.transform_net
        return (isinstance(net, ResidualNet) and not net.records_grad(inputs)
                and self.tails in (None, 'linear')
                and _lib.made_rqs_supported(self.num_bins, self.tails, net.hidden_features))

    def _run(self, inputs, context, inverse, want_bins=False):
        if not self._fusable(inputs):
            return super()._run(inputs, context, inverse)
        _check_2d(inputs, self.features)
        if not inputs.is_contiguous():
            inputs = inputs.contiguous()
        net = self.transform_net
        if (FUSED_COUPLING_LAYER and net.hidden_stack_fusable(inputs, context)
                and _lib.coupling_layer_supported(self.features, self.num_identity_features, net.hidden_features,
                                                  self.num_transform_features, self.num_bins, self.tails)):
            # the whole layer in one launch (nf_coupling_rqs_layer_fwd): rows in, hidden stack + final Linear + spline
            # in LDS, rows out - exact fp32 products whatever the matrix mode
            outputs, logabsdet, bins = _lib.coupling_rqs_layer(
                inputs, self._index32('identity_features'), self._feat_index(), net._hidden_blob(), net.hidden_features,
                len(net.blocks), self._packed_coupling_final(), self.tail_bound, inverse=inverse,
                min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                min_derivative=self.min_derivative, want_bins=want_bins)
            return (outputs, logabsdet, bins) if want_bins else (outputs, logabsdet)
        if net.hidden_stack_fusable(inputs, context):
            # the hidden-stack launch reads the identity columns anyway: it also drops them into the output, the
            # fused launch below writes the transformed columns - every column written once, no clone of the input
            outputs = torch.empty_like(inputs)
            hidden = net.forward_hidden_columns(inputs, self._index32('identity_features'), copy_out=outputs)
        else:
            hidden = net.forward_hidden(inputs[:, self.identity_features].contiguous(), context)
            outputs = inputs.clone()                   # identity columns pass through
        pf = net.final_layer._packed_fused_final(self.num_transform_features, self.num_bins, self.tails,
                                                 logit_div=self._pre_div())
        _, logabsdet, bins = _lib.made_rqs(hidden, pf, inputs, self.tail_bound, inverse=inverse, out=outputs,
                                           min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                                           min_derivative=self.min_derivative, feat_index=self._feat_index(),
                                           want_bins=want_bins)
        if want_bins:
            return outputs, logabsdet, bins
        return outputs, logabsdet

    def _run_autograd(self, inputs, context, inverse):
        """Training path, either direction: the transformed columns are gathered (torch ops, differentiable), the
        spline and its backward run as RqsFn (nf_rqs_fwd / nf_rqs_bwd / nf_rqs_inverse_bwd with the coupling
        pre-divisor), and the result is scattered back beside the identity columns."""
        from .autograd import RqsFn
        if self.tails not in (None, 'linear'):
            raise RuntimeError('{} tails are not implemented.'.format(self.tails))
        identity_split = inputs[:, self.identity_features].contiguous()
        params = self.transform_net(identity_split, context).contiguous()
        xt = inputs.index_select(1, self.transform_features).contiguous()
        yt, logabsdet = RqsFn.apply(xt, params, self.num_bins, self.tails, self.tail_bound,
                                    (self.min_bin_width, self.min_bin_height, self.min_derivative),
                                    float(self._pre_div()), bool(inverse))
        return inputs.index_copy(1, self.transform_features, yt), logabsdet

    def _coupling_elementwise(self, inputs, transform_params, outputs, logabsdet, inverse):
        pre_div = self._pre_div()
        if self.tails not in (None, 'linear'):
            raise RuntimeError('{} tails are not implemented.'.format(self.tails))
        _lib.rqs(inputs, transform_params, self.num_bins, self.tails, self.tail_bound, inverse=inverse,
                 pre_div=pre_div, feat_index=self._feat_index(), out=outputs, logabsdet=logabsdet,
                 min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                 min_derivative=self.min_derivative, num_features=self.num_transform_features)


class PiecewiseRationalQuadraticCDF(nn.Module):
    """Elementwise RQ spline with LEARNED, batch-independent parameters (reference transforms.py:249-318): what
    `apply_unconditional_transform=True` puts on a coupling layer's identity half.  `shape` = [features] (2-D inputs).
    Runs on the spline kernel (nf_rqs_fwd / nf_rqs_bwd) with the parameter row shared across the batch."""

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1., identity_init=False,
                 min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=DEFAULT_MIN_DERIVATIVE):
        super().__init__()
        shape = list(shape)
        if len(shape) != 1:
            raise NotImplementedError('image (4-D) inputs are outside this path')
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tail_bound = tail_bound
        self.tails = tails
        num_derivatives = (num_bins - 1) if tails == 'linear' else (num_bins + 1)
        if identity_init:
            self.unnormalized_widths = nn.Parameter(torch.zeros(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.zeros(*shape, num_bins))
            constant = np.log(np.exp(1 - min_derivative) - 1)
            self.unnormalized_derivatives = nn.Parameter(constant * torch.ones(*shape, num_derivatives))
        else:
            self.unnormalized_widths = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_derivatives = nn.Parameter(torch.rand(*shape, num_derivatives))

    def _spline(self, inputs, inverse=False):
        if self.tails not in (None, 'linear'):
            raise RuntimeError('{} tails are not implemented.'.format(self.tails))
        _check_2d(inputs, self.unnormalized_widths.shape[0])
        B = inputs.shape[0]
        inputs = inputs.contiguous()
        row = torch.cat([self.unnormalized_widths, self.unnormalized_heights, self.unnormalized_derivatives], dim=-1)
        params = row.reshape(1, -1).expand(B, -1).contiguous()       # the kernel reads one parameter row per sample
        mins = (self.min_bin_width, self.min_bin_height, self.min_derivative)
        if torch.is_grad_enabled() and (inputs.requires_grad or params.requires_grad):
            from .autograd import RqsFn
            return RqsFn.apply(inputs, params, self.num_bins, self.tails, self.tail_bound, mins, 1.0, bool(inverse))
        y, lad, _ = _lib.rqs(inputs, params, self.num_bins, self.tails, self.tail_bound, inverse=inverse,
                             min_bin_width=mins[0], min_bin_height=mins[1], min_derivative=mins[2])
        return y, lad

    def forward(self, inputs, context=None):
        return self._spline(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, inverse=True)


class PiecewiseCubicCouplingTransform(CouplingTransform):
    """Cubic-spline coupling layer (reference transforms.py:349-415): `tails=None` -> the spline's default interval
    [-10, 10], 'linear' -> identity outside [-tail_bound, tail_bound]; width / height logits divided by
    sqrt(hidden_features) when the conditioner exposes it."""

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.,
                 apply_unconditional_transform=False, img_shape=None, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=DEFAULT_MIN_BIN_HEIGHT):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.tails = tails
        self.tail_bound = tail_bound
        if img_shape:
            raise NotImplementedError('image (4-D) inputs are outside this path')
        unconditional_transform = None
        if apply_unconditional_transform:
            unconditional_transform = lambda features: PiecewiseCubicCDF(       # noqa: E731
                shape=[features], num_bins=num_bins, tails=tails, tail_bound=tail_bound, min_bin_width=min_bin_width,
                min_bin_height=min_bin_height)
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)

    def _transform_dim_multiplier(self):
        return self.num_bins * 2 + 2

    def _run_autograd(self, inputs, context, inverse):
        """Training path, either direction: transformed columns gathered, cubic spline + its backward kernel as
        FamilySplineFn (nf_cubic_spline_fwd / _bwd with the coupling pre-divisor), scattered back."""
        from .autograd import FamilySplineFn
        if self.tails not in (None, 'linear'):
            raise RuntimeError('{} tails are not implemented.'.format(self.tails))
        pre_div = np.sqrt(self.transform_net.hidden_features) if hasattr(self.transform_net, 'hidden_features') else 1.0
        identity_split = inputs[:, self.identity_features].contiguous()
        params = self.transform_net(identity_split, context).contiguous()
        xt = inputs.index_select(1, self.transform_features).contiguous()
        cfg = dict(num_bins=self.num_bins, tails=self.tails, tail_bound=10.0 if self.tails is None else self.tail_bound,
                   pre_div=float(pre_div), min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height)
        yt, logabsdet = FamilySplineFn.apply(xt, params, 'cubic', cfg, bool(inverse))
        return inputs.index_copy(1, self.transform_features, yt), logabsdet

    def _coupling_elementwise(self, inputs, transform_params, outputs, logabsdet, inverse):
        pre_div = np.sqrt(self.transform_net.hidden_features) if hasattr(self.transform_net, 'hidden_features') else 1.0
        if self.tails not in (None, 'linear'):
            raise RuntimeError('{} tails are not implemented.'.format(self.tails))
        _lib.cubic_spline(inputs, transform_params, self.num_bins, tails=self.tails,
                          bound=10.0 if self.tails is None else self.tail_bound, inverse=inverse, pre_div=pre_div,
                          feat_index=self._feat_index(), out=outputs, logabsdet=logabsdet,
                          min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                          num_features=self.num_transform_features)


class PiecewiseCubicCDF(nn.Module):
    """Elementwise cubic spline with LEARNED, batch-independent parameters (reference transforms.py:417-473): what
    `apply_unconditional_transform=True` puts on a cubic coupling layer's identity half.  `shape` = [features].  With
    tails=None the reference's class calls the dangling name `splines.cubic_spline` (AttributeError there); here it
    is the spline on its default interval [-10, 10] (cubic.py:56-67), like the autoregressive cubic transform."""

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1., min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=DEFAULT_MIN_BIN_HEIGHT):
        super().__init__()
        shape = list(shape)
        if len(shape) != 1:
            raise NotImplementedError('image (4-D) inputs are outside this path')
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.tail_bound = tail_bound
        self.tails = tails
        self.unnormalized_widths = nn.Parameter(torch.randn(*shape, num_bins))
        self.unnormalized_heights = nn.Parameter(torch.randn(*shape, num_bins))
        self.unnorm_derivatives_left = nn.Parameter(torch.randn(*shape, 1))
        self.unnorm_derivatives_right = nn.Parameter(torch.randn(*shape, 1))

    def _spline(self, inputs, inverse=False):
        if self.tails not in (None, 'linear'):
            raise RuntimeError('{} tails are not implemented.'.format(self.tails))
        _check_2d(inputs, self.unnormalized_widths.shape[0])
        B = inputs.shape[0]
        inputs = inputs.contiguous()
        row = torch.cat([self.unnormalized_widths, self.unnormalized_heights, self.unnorm_derivatives_left,
                         self.unnorm_derivatives_right], dim=-1)
        params = row.reshape(1, -1).expand(B, -1).contiguous()       # the kernel reads one parameter row per sample
        cfg = dict(num_bins=self.num_bins, tails=self.tails, tail_bound=10.0 if self.tails is None else self.tail_bound,
                   min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height)
        if torch.is_grad_enabled() and (inputs.requires_grad or params.requires_grad):
            from .autograd import FamilySplineFn
            return FamilySplineFn.apply(inputs, params, 'cubic', cfg, bool(inverse))
        y, lad, _ = _lib.family_spline('cubic', inputs, params, inverse=inverse, **cfg)
        return y, lad

    def forward(self, inputs, context=None):
        return self._spline(inputs, inverse=False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, inverse=True)
