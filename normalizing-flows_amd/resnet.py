"""Dense residual MLP: the coupling-layer conditioner (reference models/nn/resnet.py:9-100).

Same constructor, submodule names and `hidden_features` attribute as the reference; each layer is
the fused MFMA kernel of ``made.py`` with no mask (ReLU / eval-BatchNorm prologue, bias + skip
epilogue inside the kernel).
"""
import torch
from torch import nn
from torch.nn import functional as F, init

from . import _lib
from .made import DenseLinear, _bn_scale_shift, _bn_through_torch, _check_activation, _is_relu


class ResidualBlock(nn.Module):
    """x + L1(relu(BN1?(L0(relu(BN0?(x)))))) (reference resnet.py:9-57)."""

    def __init__(self, features, context_features, activation=F.relu, dropout_probability=0.,
                 use_batch_norm=False, zero_initialization=True):
        super().__init__()
        _check_activation(activation)
        self.activation = activation
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList([nn.BatchNorm1d(features, eps=1e-3) for _ in range(2)])
        if context_features is not None:
            self.context_layer = DenseLinear(context_features, features)
        self.linear_layers = nn.ModuleList([DenseLinear(features, features) for _ in range(2)])
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)
            init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        s0 = h0 = s1 = h1 = None
        if not _is_relu(self.activation):
            # general activation (resnet.py:38-57 op for op): the callable between the layers' launches
            temps = self.batch_norm_layers[0](inputs) if self.use_batch_norm else inputs
            temps = self.linear_layers[0].fused(self.activation(temps))
            if self.use_batch_norm:
                temps = self.batch_norm_layers[1](temps)
            temps = self.dropout(self.activation(temps))
            if context is None:
                return self.linear_layers[1].fused(temps, residual=inputs)
            return inputs + self.linear_layers[1].fused(temps) * torch.sigmoid(self.context_layer(context))
        if self.training and self.dropout.p > 0:
            # drop(act(BN1?(.))) in train() mode (resnet.py:44-50): torch's dropout between the two layers
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
            temps = self.linear_layers[0].fused(inputs, in_scale=s0, in_shift=h0, relu_in=True)
        if context is None:
            return self.linear_layers[1].fused(temps, in_scale=s1, in_shift=h1, relu_in=True, residual=inputs)
        temps = self.linear_layers[1].fused(temps, in_scale=s1, in_shift=h1, relu_in=True)
        return inputs + temps * torch.sigmoid(self.context_layer(context))


class ResidualNet(nn.Module):
    """initial linear -> residual blocks -> final linear (reference resnet.py:60-100)."""

    def __init__(self, in_features, out_features, hidden_features, context_features=None, num_blocks=2,
                 activation=F.relu, dropout_probability=0., use_batch_norm=False):
        super().__init__()
        self.hidden_features = hidden_features
        self.context_features = context_features
        first_in = in_features if context_features is None else in_features + context_features
        self.initial_layer = DenseLinear(first_in, hidden_features)
        self.blocks = nn.ModuleList([
            ResidualBlock(features=hidden_features, context_features=context_features, activation=activation,
                          dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
            for _ in range(num_blocks)])
        self.final_layer = DenseLinear(hidden_features, out_features)

    def _hidden_blob(self):
        """Parameters of the hidden stack in the layout of nf_resnet_blob_floats, rebuilt when any changes."""
        tensors = [self.initial_layer.weight, self.initial_layer.bias]
        for blk in self.blocks:
            for j in range(2):
                if blk.use_batch_norm:
                    bn = blk.batch_norm_layers[j]
                    tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
                tensors += [blk.linear_layers[j].weight, blk.linear_layers[j].bias]
        key = tuple((t.data_ptr(), _lib.weight_version(t)) for t in tensors) + (self.initial_layer.weight.device,)
        cached = getattr(self, '_blob_cache', None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                w0 = self.initial_layer.weight
                kp = (w0.shape[1] + 7) // 8 * 8
                parts = [F.pad(w0, (0, kp - w0.shape[1])).reshape(-1), self.initial_layer.bias]
                for blk in self.blocks:
                    for j in range(2):
                        if blk.use_batch_norm:
                            sc, sh = _bn_scale_shift(blk.batch_norm_layers[j])
                        else:
                            sc, sh = torch.ones_like(self.initial_layer.bias), torch.zeros_like(self.initial_layer.bias)
                        parts += [sc, sh, blk.linear_layers[j].weight.reshape(-1), blk.linear_layers[j].bias]
                blob = torch.cat([p.detach().float().reshape(-1) for p in parts]).contiguous()
            object.__setattr__(self, '_blob_cache', (key, blob))
            cached = self._blob_cache
        return cached[1]

    def records_grad(self, inputs):
        """True when autograd would record this call (the inference-only fused kernels must not run then)."""
        return torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))

    def hidden_stack_fusable(self, inputs, context=None):
        """One launch for initial layer + blocks (nf_resnet_hidden_fwd): inference, no context, ReLU,
        hidden width a multiple of 32 up to 128."""
        return (context is None and self.context_features is None and not self.records_grad(inputs)
                and not any(b.training and (b.use_batch_norm or b.dropout.p > 0) for b in self.blocks)
                and all(_is_relu(b.activation) for b in self.blocks)
                and _lib.resnet_hidden_supported(self.initial_layer.in_features, self.hidden_features))

    def forward_hidden_columns(self, inputs, columns, copy_out=None):
        """forward_hidden(inputs[:, columns]) without materialising the gather; `columns` int32 on the device.
        copy_out: receives those columns of `inputs` at the same positions (the coupling layer's identity half)."""
        return _lib.resnet_hidden(inputs, columns, self._hidden_blob(), self.initial_layer.in_features,
                                  self.hidden_features, len(self.blocks), copy_out=copy_out)

    def forward_hidden(self, inputs, context=None):
        """Everything but the final layer (which the coupling transforms fuse with their spline)."""
        if context is not None:
            inputs = torch.cat((inputs, context), dim=1)
        temps = self.initial_layer(inputs)
        for block in self.blocks:
            temps = block(temps, context=context)
        return temps

    def forward(self, inputs, context=None):
        return self.final_layer(self.forward_hidden(inputs, context))
