"""MADE masked-linear conditioner on the MI355X matrix cores.

Drop-in for the reference's models/made.py:91-359 on this path: same constructor arguments, same
submodule / buffer names (so a reference ``state_dict`` loads with ``strict=True``), same
``forward(inputs, context=None) -> [B, features*output_multiplier]``.  What differs is the execution:
``weight * mask`` is folded once into a packed, tile-padded device matrix with a table of unmasked K
tiles (nf_made_pack) and every layer runs as one fused MFMA kernel (nf_masked_linear_fwd) with the
ReLU / eval-BatchNorm prologue and the bias / skip epilogue inside it.

Activations: ReLU (the reference default) is fused into the kernels; any other callable runs between the launches
(_is_relu).  Inference (``eval()`` / no-grad) runs entirely on
the fused kernels; under autograd the GEMMs run as ``MaskedLinearFn`` (forward / input-gradient / weight-gradient
kernels) and BatchNorm (train() statistics), dropout and the context GLU gate are torch ops between them.
Anything else raises instead of silently taking another path.
"""
import torch
from torch import nn
from torch.nn import functional as F, init

from . import _lib

_MATMUL_DTYPE = {'f32': _lib.NF_DTYPE_F32, 'bf16': _lib.NF_DTYPE_BF16, 'bf16x3': _lib.NF_DTYPE_BF16X3,
                 'f16x2': _lib.NF_DTYPE_F16X2}
# modes whose hidden-layer GEMMs run on the split-operand kernel (nf_masked_linear_split_fwd): the two split modes, and
# 'bf16' as that kernel's one-term flavour (plain bf16 operands, one MFMA per product)
_SPLIT_MODES = ('bf16x3', 'f16x2', 'bf16')
# hidden-layer GEMMs the split kernel does not take (BatchNorm prologue, tile windows, ragged K) stay on fp32 MFMA
_GEMM_DTYPE = {'f32': _lib.NF_DTYPE_F32, 'bf16': _lib.NF_DTYPE_BF16, 'bf16x3': _lib.NF_DTYPE_F32,
               'f16x2': _lib.NF_DTYPE_F32}
_default_matmul = 'f32'


def set_matmul_dtype(name):
    """'f32' (exact fp32 MFMA, the 1e-5 contract), 'bf16' (bf16 MFMA, the 2e-2 contract), 'bf16x3' (the
    fused final layer on six bf16 MFMAs per product of three-term split fp32 operands: the 1e-5 contract) or
    'f16x2' (three fp16 MFMAs per product of two-term split operands; weights scaled by a power of two at pack time,
    the hidden state by one found per launch from its largest magnitude, so any finite range is taken: the 1e-5
    contract holds for every element within 2^-14 of the launch's largest |h|; smaller ones keep 36 - log2(max/|h|)
    bits - a hidden state with a 1e6 outlier unit still gives 1e-4, tests/test_gpu_parity.py::
    test_f16x2_with_a_heavy_tailed_hidden_state).  Whole-layer inference launches of the autoregressive RQ transform
    (transforms.FUSED_HIDDEN_STACK: hidden stack + final layer as two launches) scale every group of 64 ROWS by its
    own power of two instead - a batch whose samples differ in size by orders of magnitude costs the small ones
    nothing there (tests/test_gpu_parity.py::test_rq_ar_layer_with_and_without_the_fused_hidden_stack); the bound above
    then applies within a 64-row group only."""
    global _default_matmul
    if name not in _MATMUL_DTYPE:
        raise ValueError('matmul dtype must be one of %s' % sorted(_MATMUL_DTYPE))
    _default_matmul = name


def get_matmul_dtype():
    return _default_matmul


def tile(x, n):
    """Each element repeated n times in place order (reference helper_functions.py:382-390)."""
    if not (isinstance(n, int) and n > 0):
        raise TypeError('Argument \'n\' must be a positive integer.')
    return x.reshape(-1, 1).repeat(1, n).reshape(-1)


def _get_input_degrees(in_features):
    return torch.arange(1, in_features + 1)


def _training(module):
    """True when autograd should record this call (grad mode on and the layer has trainable parameters)."""
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters(recurse=False))


def _require_inference(module, activation=None):
    if _training(module):
        raise NotImplementedError(
            'this configuration has no backward kernel: call under torch.no_grad() (trainable today: '
            'RQ-spline / affine autoregressive and coupling transforms, forward direction)')


class _PackedMixin:
    """Caches the packed weights; rebuilt when the parameters change or the MFMA dtype is switched."""

    def _packed_linear(self):
        mask = getattr(self, 'mask', None)
        mkey = None if mask is None else (mask._version, mask.data_ptr())
        key = (_lib.weight_version(self.weight), self.weight.data_ptr(), None if self.bias is None else _lib.weight_version(self.bias),
               get_matmul_dtype(), self.weight.device, mkey)
        cached = getattr(self, '_pack_cache', None)
        if cached is None or cached[0] != key:
            same_mask = cached is not None and cached[0][-1] == mkey      # then the K-tile table is still valid
            pl = _lib.PackedLinear(self.weight, mask, self.bias, _GEMM_DTYPE[get_matmul_dtype()],
                                   reuse=cached[1] if same_mask else None)
            object.__setattr__(self, '_pack_cache', (key, pl))
            cached = self._pack_cache
        return cached[1]

    def _packed_linear_split(self):
        """Term-plane pack for the split-operand hidden-layer kernel (nf_linear_split_pack), cached likewise."""
        mask = getattr(self, 'mask', None)
        key = (_lib.weight_version(self.weight), self.weight.data_ptr(), None if self.bias is None else _lib.weight_version(self.bias),
               get_matmul_dtype(), self.weight.device, None if mask is None else (mask._version, mask.data_ptr()))
        cached = getattr(self, '_split_cache', None)
        if cached is None or cached[0] != key:
            same_mask = cached is not None and cached[0][-1] == key[-1]       # then the K-range table is still valid
            pls = _lib.PackedLinearSplit(self.weight, mask, self.bias, _MATMUL_DTYPE[get_matmul_dtype()],
                                         reuse=cached[1] if same_mask else None)
            object.__setattr__(self, '_split_cache', (key, pls))
            cached = self._split_cache
        return cached[1]

    def _packed_linear_split_t(self):
        """Term-plane pack of the transposed layer (input-gradient GEMM in the split-operand modes), cached likewise."""
        mask = getattr(self, 'mask', None)
        key = (_lib.weight_version(self.weight), self.weight.data_ptr(), get_matmul_dtype(), self.weight.device,
               None if mask is None else (mask._version, mask.data_ptr()))
        cached = getattr(self, '_split_t_cache', None)
        if cached is None or cached[0] != key:
            same_mask = cached is not None and cached[0][-1] == key[-1]
            pls = _lib.PackedLinearSplit(self.weight, mask, None, _MATMUL_DTYPE[get_matmul_dtype()],
                                         reuse=cached[1] if same_mask else None, transposed=True)
            object.__setattr__(self, '_split_t_cache', (key, pls))
            cached = self._split_t_cache
        return cached[1]

    def _wgrad_ktab(self):
        """K-tile table of the 128-row output tiles (nf_made_pack's) for the weight-gradient kernels: a function of the
        mask only, so it is built once and kept across optimizer steps."""
        mask = getattr(self, 'mask', None)
        key = (None if mask is None else (mask._version, mask.data_ptr()), self.weight.device, tuple(self.weight.shape))
        cached = getattr(self, '_ktab_cache', None)
        if cached is None or cached[0] != key:
            cached = (key, _lib.PackedLinear(self.weight, mask, None, _lib.NF_DTYPE_F32).ktab)
            object.__setattr__(self, '_ktab_cache', cached)
        return cached[1]

    def _packed_fused_final(self, features, num_bins, tails, logit_div=1.0):
        """Final layer regrouped for the fused GEMM + spline kernel (nf_made_rqs_pack), cached likewise."""
        mask = getattr(self, 'mask', None)
        mkey = None if mask is None else (mask._version, mask.data_ptr())
        key = (_lib.weight_version(self.weight), self.weight.data_ptr(), None if self.bias is None else _lib.weight_version(self.bias),
               get_matmul_dtype(), self.weight.device, features, num_bins, tails, float(logit_div), mkey)
        cached = getattr(self, '_fused_cache', None)
        if cached is None or cached[0] != key:
            same = cached is not None and cached[0][3:] == key[3:]       # same mode, shape and mask: rewrite in place
            pf = _lib.PackedFusedFinal(self.weight, mask, self.bias, features, num_bins, tails,
                                       _MATMUL_DTYPE[get_matmul_dtype()], logit_div=logit_div,
                                       reuse=cached[1] if same else None)
            object.__setattr__(self, '_fused_cache', (key, pf))
            cached = self._fused_cache
        return cached[1]

    def _packed_linear_t(self):
        """Transposed pack for the input-gradient GEMM, cached like the forward pack."""
        mask = getattr(self, 'mask', None)
        mkey = None if mask is None else (mask._version, mask.data_ptr())
        dtype = _GEMM_DTYPE[get_matmul_dtype()]         # bf16 mode: the input-gradient GEMM runs on bf16 MFMA too
        key = (_lib.weight_version(self.weight), self.weight.data_ptr(), self.weight.device, dtype, mkey)
        cached = getattr(self, '_pack_t_cache', None)
        if cached is None or cached[0] != key:
            same_mask = cached is not None and cached[0][-1] == mkey
            plt = _lib.PackedLinearT(self.weight, mask, dtype, reuse=cached[1] if same_mask else None)
            object.__setattr__(self, '_pack_t_cache', (key, plt))
            cached = self._pack_t_cache
        return cached[1]

    def fused(self, x, residual=None, in_scale=None, in_shift=None, relu_in=False, relu_out=False, out=None,
              tiles=None):
        if _training(self) or (torch.is_grad_enabled() and (x.requires_grad or
                                                            (residual is not None and residual.requires_grad))):
            if in_scale is not None or relu_out or out is not None or tiles is not None:
                _require_inference(self)
                raise NotImplementedError('no backward kernel for BatchNorm / feed-forward blocks')
            from .autograd import MaskedLinearFn
            return MaskedLinearFn.apply(x, self.weight, self.bias, residual, self, bool(relu_in))
        if (get_matmul_dtype() in _SPLIT_MODES and in_scale is None and in_shift is None and tiles is None
                and _lib.split_gemm_supported(x, self.in_features)):
            return _lib.masked_linear_split(x, self._packed_linear_split(), residual=residual, relu_in=relu_in,
                                            relu_out=relu_out, out=out)
        return _lib.masked_linear(x, self._packed_linear(), residual=residual, in_scale=in_scale,
                                  in_shift=in_shift, relu_in=relu_in, relu_out=relu_out, out=out, tiles=tiles)

    def tiles_of_degree(self, degree):
        """(first, count) of the 128-column output tiles that hold a unit of this MADE degree, or None."""
        cache = getattr(self, '_degree_tiles', None)
        if cache is None:
            cache = {}
            deg = self.degrees.cpu().tolist()
            for unit, d in enumerate(deg):
                t = unit // 128
                lo, hi = cache.get(d, (t, t))
                cache[d] = (min(lo, t), max(hi, t))
            object.__setattr__(self, '_degree_tiles', cache)
        span = cache.get(int(degree))
        return None if span is None else (span[0], span[1] - span[0] + 1)


class DenseLinear(_PackedMixin, nn.Linear):
    """nn.Linear executed by the same kernel with no mask (context layers, ResidualNet)."""

    def forward(self, x):
        return self.fused(x)


class MaskedLinear(_PackedMixin, nn.Linear):
    """Linear layer with a fixed autoregressive 0/1 mask (reference made.py:96-152)."""

    def __init__(self, in_degrees, out_features, autoregressive_features, random_mask, is_output, bias=True):
        super().__init__(in_features=len(in_degrees), out_features=out_features, bias=bias)
        mask, degrees = self._get_mask_and_degrees(in_degrees, out_features, autoregressive_features,
                                                   random_mask, is_output)
        self.register_buffer('mask', mask)
        self.register_buffer('degrees', degrees)

    @classmethod
    def _get_mask_and_degrees(cls, in_degrees, out_features, autoregressive_features, random_mask, is_output):
        in_degrees = in_degrees.cpu()
        if is_output:
            # every feature's parameters may see strictly earlier features only
            out_degrees = tile(_get_input_degrees(autoregressive_features), out_features // autoregressive_features)
            mask = (out_degrees[..., None] > in_degrees).float()
        else:
            if random_mask:
                low = min(torch.min(in_degrees).item(), autoregressive_features - 1)
                out_degrees = torch.randint(low=low, high=autoregressive_features, size=[out_features],
                                            dtype=torch.long)
            else:
                hi = max(1, autoregressive_features - 1)
                lo = min(1, autoregressive_features - 1)
                out_degrees = torch.arange(out_features) % hi + lo
            mask = (out_degrees[..., None] >= in_degrees).float()
        return mask, out_degrees

    def forward(self, x):
        return self.fused(x)


def _bn_scale_shift(bn):
    """Eval-mode BatchNorm1d as y = x*scale + shift."""
    if bn.training:
        raise NotImplementedError('BatchNorm in train() mode couples samples across the batch; call eval().')
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    return scale.contiguous(), shift.contiguous()


def _records(module, inputs):
    return torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in module.parameters()))


def _bn_through_torch(block, inputs):
    """BatchNorm runs as torch's own op (not folded into the next GEMM's input) in train() mode - batch statistics
    are a cross-sample reduction and the running statistics get updated, as in the reference - and whenever
    autograd records; the layers around it stay on the kernels."""
    return block.training or _records(block, inputs)


def _is_relu(activation):
    """The reference default (F.relu) is fused into the GEMM kernels (prologue / epilogue flags); any other callable
    (`activation=torch.tanh`, F.elu, ...: made.py:302, resnet.py:16) runs as itself - an elementwise op on the device
    tensor - between the layers' launches, and the ReLU-only fast paths (row-resident / windowed inverse, one-launch
    coupling conditioner) step aside for the general ones."""
    return activation is F.relu or activation is torch.relu


def _check_activation(activation):
    if not callable(activation):
        raise TypeError('activation must be callable')


class MaskedFeedforwardBlock(nn.Module):
    """BN? -> masked linear -> relu -> dropout (reference made.py:155-207)."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0., use_batch_norm=False):
        super().__init__()
        features = len(in_degrees)
        self.batch_norm = nn.BatchNorm1d(features, eps=1e-3) if use_batch_norm else None
        if context_features is not None:
            raise NotImplementedError()
        self.linear = MaskedLinear(in_degrees, features, autoregressive_features, random_mask, is_output=False)
        self.degrees = self.linear.degrees
        _check_activation(activation)
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)

    def forward(self, inputs, context=None):
        if context is not None:
            raise NotImplementedError()
        dropping = self.training and self.dropout.p > 0
        if not _is_relu(self.activation) or dropping or _records(self, inputs) or \
                (self.batch_norm is not None and self.training):
            temps = inputs if self.batch_norm is None else self.batch_norm(inputs)
            return self.dropout(self.activation(self.linear.fused(temps)))
        sc, sh = _bn_scale_shift(self.batch_norm) if self.batch_norm is not None else (None, None)
        return self.linear.fused(inputs, in_scale=sc, in_shift=sh, relu_out=True)


class MaskedResidualBlock(nn.Module):
    """x + ML1(relu(BN1?(ML0(relu(BN0?(x)))))) (reference made.py:210-284)."""

    def __init__(self, in_degrees, autoregressive_features, context_features=None, random_mask=False,
                 activation=F.relu, dropout_probability=0., use_batch_norm=False, zero_initialization=True):
        if random_mask:
            raise ValueError('Masked residual block can\'t be used with random masks.')
        super().__init__()
        features = len(in_degrees)
        if context_features is not None:
            self.context_layer = DenseLinear(context_features, features)
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList([nn.BatchNorm1d(features, eps=1e-3) for _ in range(2)])
        linear_0 = MaskedLinear(in_degrees, features, autoregressive_features, False, is_output=False)
        linear_1 = MaskedLinear(linear_0.degrees, features, autoregressive_features, False, is_output=False)
        self.linear_layers = nn.ModuleList([linear_0, linear_1])
        self.degrees = linear_1.degrees
        if torch.all(self.degrees >= in_degrees.cpu()).item() != 1:
            raise RuntimeError('In a masked residual block, the output degrees can\'t be'
                               ' less than the corresponding input degrees.')
        _check_activation(activation)
        self.activation = activation
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            init.uniform_(self.linear_layers[-1].weight, a=-1e-3, b=1e-3)
            init.uniform_(self.linear_layers[-1].bias, a=-1e-3, b=1e-3)

    def forward(self, inputs, context=None):
        s0 = h0 = s1 = h1 = None
        if not _is_relu(self.activation):
            # general activation (made.py:268-284 op for op): the callable between the layers' launches
            temps = self.batch_norm_layers[0](inputs) if self.use_batch_norm else inputs
            temps = self.linear_layers[0].fused(self.activation(temps))
            if self.use_batch_norm:
                temps = self.batch_norm_layers[1](temps)
            temps = self.dropout(self.activation(temps))
            if context is None:
                return self.linear_layers[1].fused(temps, residual=inputs)
            return inputs + self.linear_layers[1].fused(temps) * torch.sigmoid(self.context_layer(context))
        if self.training and self.dropout.p > 0:
            # drop(act(BN1?(.))) in train() mode (made.py:275-278): torch's dropout between the two layers
            bn = self.batch_norm_layers if self.use_batch_norm else (lambda t: t, lambda t: t)
            temps = bn[1](self.linear_layers[0].fused(bn[0](inputs), relu_in=True))
            temps = self.dropout(torch.relu(temps))
            if context is None:
                return self.linear_layers[1].fused(temps, residual=inputs)
            return inputs + self.linear_layers[1].fused(temps) * torch.sigmoid(self.context_layer(context))
        if self.use_batch_norm and _bn_through_torch(self, inputs):
            temps = self.linear_layers[0].fused(self.batch_norm_layers[0](inputs), relu_in=True)
            temps = self.batch_norm_layers[1](temps)
        else:
            if self.use_batch_norm:
                s0, h0 = _bn_scale_shift(self.batch_norm_layers[0])
                s1, h1 = _bn_scale_shift(self.batch_norm_layers[1])
            elif context is None and (_training(self.linear_layers[0]) or _training(self.linear_layers[1]) or
                                      (torch.is_grad_enabled() and inputs.requires_grad)):
                from . import autograd as _ag
                if _ag.FUSED_RESIDUAL_BLOCK:            # the whole block as one autograd node
                    l0, l1 = self.linear_layers
                    return _ag.ResidualBlockFn.apply(inputs, l0.weight, l0.bias, l1.weight, l1.bias, l0, l1)
            temps = self.linear_layers[0].fused(inputs, in_scale=s0, in_shift=h0, relu_in=True)
        if context is None:
            return self.linear_layers[1].fused(temps, in_scale=s1, in_shift=h1, relu_in=True, residual=inputs)
        temps = self.linear_layers[1].fused(temps, in_scale=s1, in_shift=h1, relu_in=True)
        # GLU gate from the context (made.py:279-283): elementwise glue around the fused layers
        return inputs + temps * torch.sigmoid(self.context_layer(context))


class MADE(nn.Module):
    """Masked autoregressive conditioner (reference made.py:287-359)."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2, output_multiplier=1,
                 use_residual_blocks=True, random_mask=False, activation=F.relu, dropout_probability=0.,
                 use_batch_norm=False):
        if use_residual_blocks and random_mask:
            raise ValueError('Residual blocks can\'t be used with random masks.')
        super().__init__()
        self.initial_layer = MaskedLinear(_get_input_degrees(features), hidden_features, features, random_mask,
                                          is_output=False)
        if context_features is not None:
            self.context_layer = DenseLinear(context_features, hidden_features)
        block = MaskedResidualBlock if use_residual_blocks else MaskedFeedforwardBlock
        prev = self.initial_layer.degrees
        blocks = []
        for _ in range(num_blocks):
            blocks.append(block(in_degrees=prev, autoregressive_features=features,
                                context_features=context_features, random_mask=random_mask,
                                activation=activation, dropout_probability=dropout_probability,
                                use_batch_norm=use_batch_norm))
            prev = blocks[-1].degrees
        self.blocks = nn.ModuleList(blocks)
        self.final_layer = MaskedLinear(prev, features * output_multiplier, features, random_mask, is_output=True)

    def forward_hidden(self, inputs, context=None):
        """Everything up to (not including) the final layer: [B, hidden_features]."""
        if context is None:
            outputs = self.initial_layer.fused(inputs)
        else:
            outputs = self.initial_layer.fused(inputs, residual=self.context_layer(context))
        for block in self.blocks:
            outputs = block(outputs, context)
        return outputs

    # ---- incremental evaluation (autoregressive inverse) --------------------------------------------------
    def hidden_program(self):
        """The hidden stack as a list of fused-layer steps over named buffers, for evaluating it
        tile window by tile window: (layer, src, dst, residual or None, relu_in, relu_out, bn or None).
        Buffer 'x' is the network input; the last dst is the hidden state the final layer reads."""
        if any(b.training and b.dropout.p > 0 for b in self.blocks):
            raise NotImplementedError('dropout is active (train() mode): the inverse kernels evaluate the network '
                                      'without it - call eval()')
        if not self.relu_only():
            raise NotImplementedError('the incremental inverse kernels fuse ReLU; other activations take the D-pass loop')
        steps = [(self.initial_layer, 'x', 'h0', None, False, False, None)]
        cur = 'h0'
        for i, block in enumerate(self.blocks):
            if isinstance(block, MaskedResidualBlock):
                bn = block.batch_norm_layers if block.use_batch_norm else (None, None)
                steps.append((block.linear_layers[0], cur, 't%d' % i, None, True, False, bn[0]))
                steps.append((block.linear_layers[1], 't%d' % i, 'h%d' % (i + 1), cur, True, False, bn[1]))
            else:
                steps.append((block.linear, cur, 'h%d' % (i + 1), None, False, True, block.batch_norm))
            cur = 'h%d' % (i + 1)
        return steps, cur

    def relu_only(self):
        return all(_is_relu(b.activation) for b in self.blocks)

    def hidden_degrees(self):
        """Sorted MADE degrees that occur on any hidden unit."""
        degs = set()
        for layer, *_ in self.hidden_program()[0]:
            degs.update(layer.degrees.cpu().tolist())
        return sorted(degs)

    def run_hidden_program(self, bufs, degree=None):
        """Runs the hidden stack into `bufs` (dict name -> [B, hidden] tensor, 'x' = input).  With `degree`
        only the output tiles that hold units of that degree are recomputed - everything a change of
        input feature `degree - 1` can alter that later features need (see transforms.py)."""
        steps, last = self.hidden_program()
        for layer, src, dst, res, relu_in, relu_out, bn in steps:
            tiles = None
            if degree is not None:
                tiles = layer.tiles_of_degree(degree)
                if tiles is None:
                    continue
            sc, sh = _bn_scale_shift(bn) if bn is not None else (None, None)
            layer.fused(bufs[src], residual=None if res is None else bufs[res], in_scale=sc, in_shift=sh,
                        relu_in=relu_in, relu_out=relu_out, out=bufs[dst], tiles=tiles)
        return bufs[last]

    def hidden_planes_pack(self):
        """Pack for the one-launch hidden stack (nf_made_hidden_planes_fwd, f16x2 arithmetic), or None when this network
        is outside its scope: ReLU, no BatchNorm, no active dropout, residual or feed-forward blocks, hidden width a
        multiple of 32 up to 512, and no first-layer unit reading an input column >= 512 (the degree-ordered masks
        with hidden_features <= 512).  A context is the caller's business: the kernel has no context path."""
        if not self.relu_only() or any(b.training and b.dropout.p > 0 for b in self.blocks):
            return None
        steps, _ = self.hidden_program()
        layers = [st[0] for st in steps]
        key = tuple((_lib.weight_version(l.weight), l.weight.data_ptr(), None if l.bias is None else _lib.weight_version(l.bias),
                     l.mask._version, l.mask.data_ptr()) for l in layers) + (self.initial_layer.weight.device,)
        cached = getattr(self, '_hidden_planes_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        pack = None
        H = self.final_layer.in_features
        types = []
        ok = all(st[6] is None for st in steps) and all(l.out_features == H for l in layers)
        for i, (layer, src, dst, res, relu_in, relu_out, bn) in enumerate(steps):
            if i == 0:
                types.append(0)
                ok = ok and not relu_in and not relu_out and res is None
            elif relu_in and not relu_out and res is None:
                types.append(1)           # first linear of a residual block
            elif relu_in and not relu_out and res is not None:
                types.append(2)
                ok = ok and types[-2] == 1 and res == steps[i - 1][1]
            elif relu_out and not relu_in and res is None:
                types.append(3)           # feed-forward block
            else:
                ok = False
                types.append(-1)
        if ok:
            live = self.initial_layer.mask.ne(0).any(0).nonzero()
            k_live0 = int(live.max().item()) + 1 if live.numel() else 1
            if _lib.made_hidden_supported(len(layers), H, k_live0):
                pack = _lib.PackedHiddenStack([(l.weight, l.mask, l.bias) for l in layers], types, k_live0)
        object.__setattr__(self, '_hidden_planes_cache', (key, pack))
        return pack

    def inverse_rows_pack(self, num_bins, tails, kind=None):
        """Parameter blob for the row-resident inverse kernel (nf_made_rqs_inverse_rows; num_bins None: the affine
        variant nf_made_affine_inverse_rows), or None when this
        network is outside its scope: hidden units must be in degree order in every layer (the non-random masks
        with hidden_features <= features - 1), BatchNorm in eval mode only, the step structure of hidden_program()."""
        kind = kind or ('affine' if num_bins is None else 'rq')
        if not self.relu_only():
            return None
        steps, _ = self.hidden_program()
        layers = [st[0] for st in steps] + [self.final_layer]
        key = (kind,) + tuple((_lib.weight_version(l.weight), l.weight.data_ptr(), None if l.bias is None else _lib.weight_version(l.bias),
                     l.mask._version) for l in layers) + (self.final_layer.weight.device, num_bins, tails)
        for st in steps:
            if st[6] is not None:
                bn = st[6]
                key += (_lib.weight_version(bn.weight), _lib.weight_version(bn.bias), bn.running_mean._version, bn.running_var._version,
                        bn.training)
        cached = getattr(self, '_inverse_rows_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        D, H = self.initial_layer.in_features, self.final_layer.in_features
        pack = None
        ok = _lib.inverse_rows_supported(kind, D, H, len(steps), num_bins or 0, tails)
        flags = []
        want = torch.arange(1, H + 1)
        for i, (layer, src, dst, res, relu_in, relu_out, bn) in enumerate(steps):
            ok = ok and (bn is None or (i > 0 and not bn.training)) and torch.equal(layer.degrees.cpu(), want)
            ok = ok and (i == 0 or src == steps[i - 1][2]) and (res is None or (i >= 2 and res == steps[i - 2][2]))
            flags.append((1 if relu_in else 0) | (2 if res is not None else 0) | (4 if relu_out else 0))
        if ok:
            pack = _lib.PackedInverse([(st[0].weight, st[0].mask, st[0].bias, fl,
                                        None if st[6] is None else _bn_scale_shift(st[6]))
                                       for st, fl in zip(steps, flags)],
                                      self.final_layer.weight, self.final_layer.mask, self.final_layer.bias, D,
                                      num_bins, tails, kind=kind)
        object.__setattr__(self, '_inverse_rows_cache', (key, pack))
        return pack

    def inverse_rows_context(self, context):
        """What the row-resident inverse kernel needs from a context: (context_layer(context), sigmoid gate of
        residual block 0, of block 1), or None without a context.  Raises for networks whose context enters any
        other way."""
        if context is None:
            return None
        if not hasattr(self, 'context_layer') or len(self.blocks) > 2 or \
                not all(isinstance(b, MaskedResidualBlock) and hasattr(b, 'context_layer') for b in self.blocks):
            raise NotImplementedError('context given to a MADE built without context_features')
        gates = [torch.sigmoid(b.context_layer(context)) for b in self.blocks]
        return (self.context_layer(context),) + tuple(gates) + (None,) * (2 - len(gates))

    def forward(self, inputs, context=None):
        return self.final_layer.fused(self.forward_hidden(inputs, context))
