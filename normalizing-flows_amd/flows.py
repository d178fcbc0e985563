"""Legacy Gaussian MAF layer (reference models/flows.py:230-263 on models/made.py:11-89), float64.

API as in the reference: ``forward(z) -> (x, logJ)``, ``reverse(x) -> (z, logJ)``; both directions train (backward
kernels nf_linear_f64_bwd / nf_maf_bwd / nf_maf_reverse_step_bwd behind autograd.Functions).  Faithful to what
the reference computes, including what SURVEY.md 2.1-4 documents: SimpleMADE's output mask uses `>=`,
so ``reverse`` is NOT the inverse of ``forward`` and ``logJ`` is not the true log-determinant; parity
means reproducing the reference's numbers.  ``reverse`` returns the logJ of the LAST pass.

The masks are plain attributes (not buffers), exactly as in the reference, and come from numpy's global
RNG (`seed=None` in MaskedAutoregressiveFlow), so they are not reproducible: transfer a reference
model with ``load_reference(state_dict, mask_matrix)``.
"""
import numpy as np
import torch
from numpy.random import permutation, randint
from torch import nn

from . import _lib


class SimpleMaskedLinear(nn.Linear):
    """nn.Linear with a mask attribute; y = x @ (mask*W)^T + b (reference made.py:11-21)."""

    def __init__(self, n_in, n_out, bias=True):
        super().__init__(n_in, n_out, bias)
        self.mask = None

    def initialise_mask(self, mask):
        self.mask = mask
        object.__setattr__(self, '_folded', None)

    def folded(self):
        key = (_lib.weight_version(self.weight), self.weight.data_ptr(), self.weight.device)
        c = getattr(self, '_folded', None)
        if c is None or c[0] != key:
            mask = None if self.mask is None else torch.as_tensor(self.mask).to(self.weight.device)
            object.__setattr__(self, '_folded', (key, _lib.fold_f64(self.weight, mask)))
        return self._folded[1]

    def mask_f64(self):
        """The mask as a float64 tensor on the weights' device (operand of the weight-gradient kernel)."""
        if self.mask is None:
            return None
        c = getattr(self, '_mask64', None)
        if c is None or c[0] is not self.mask or c[1].device != self.weight.device:
            m = torch.as_tensor(self.mask).to(device=self.weight.device, dtype=torch.float64).contiguous()
            object.__setattr__(self, '_mask64', (self.mask, m))
        return self._mask64[1]

    def forward(self, x, relu_out=False):
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad or
                                        (self.bias is not None and self.bias.requires_grad)):
            return LinearF64Fn.apply(x, self.weight, self.bias, self, bool(relu_out))
        return _lib.linear_f64(x, self.folded(), self.bias, relu_out=relu_out)


class LinearF64Fn(torch.autograd.Function):
    """y = act(x @ (mask*W)^T + b) with nf_linear_f64_fwd / nf_linear_f64_bwd (autograd through made.py:20-21)."""

    @staticmethod
    def forward(ctx, x, weight, bias, layer, relu_out):
        x = x.contiguous()
        folded = layer.folded()
        y = _lib.linear_f64(x, folded, bias, relu_out=relu_out)
        ctx.save_for_backward(x, y, folded)
        ctx.layer, ctx.relu_out, ctx.has_bias = layer, relu_out, bias is not None
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, y, folded = ctx.saved_tensors
        need = ctx.needs_input_grad
        gx, gw, gb = _lib.linear_f64_bwd(x, y, folded, ctx.layer.mask_f64(), grad_y.contiguous(), ctx.relu_out,
                                         want_input=need[0], want_weight=need[1], want_bias=ctx.has_bias and need[2])
        return gx, gw, gb, None, None


class MafFwdFn(torch.autograd.Function):
    """(x, logJ) = nf_maf_fwd(z, made_out); backward nf_maf_bwd."""

    @staticmethod
    def forward(ctx, z, made_out, reverse_input):
        z = z.contiguous()
        x, logJ = _lib.maf_fwd(z, made_out, reverse_input)
        ctx.save_for_backward(z, made_out)
        ctx.reverse_input = reverse_input
        return x, logJ

    @staticmethod
    def backward(ctx, grad_x, grad_logJ):
        z, made_out = ctx.saved_tensors
        gz, go = _lib.maf_bwd(z, made_out, None if grad_x is None else grad_x.contiguous(), grad_logJ,
                              ctx.reverse_input)
        return gz, go, None


class MafReverseStepFn(torch.autograd.Function):
    """One pass of reverse() taken out of place: z_new = z_old with column `dim` rebuilt from made_out (the
    reference writes the column in place and lets autograd version the tensor, flows.py:255-259); the last pass also
    returns logJ.  Backward: nf_maf_reverse_step_bwd."""

    @staticmethod
    def forward(ctx, x, made_out, z_old, dim, reverse_input, last):
        z_new = z_old.clone()
        logJ = torch.empty(x.shape[0], dtype=x.dtype, device=x.device) if last else None
        _lib.maf_reverse_step(x, made_out, z_new, dim, reverse_input, logJ)
        ctx.save_for_backward(x, made_out)
        ctx.cfg = (dim, reverse_input, last)
        if last:
            return z_new, logJ
        return z_new, z_new.new_zeros(())           # (placeholder: autograd.Function needs tensor outputs)

    @staticmethod
    def backward(ctx, grad_z_new, grad_logJ):
        x, made_out = ctx.saved_tensors
        dim, reverse_input, last = ctx.cfg
        gx, go, gzo = _lib.maf_reverse_step_bwd(x, made_out, None if grad_z_new is None else grad_z_new.contiguous(),
                                                grad_logJ if last else None, dim, reverse_input)
        return gx, go, gzo, None, None, None


class SimpleMADE(nn.Module):
    """Linear-ReLU stack with MADE masks from numpy's RNG (reference made.py:23-89).  Submodule layout
    `model.{0,2,4,..}` matches the reference's nn.Sequential (odd slots are the ReLUs, fused here)."""

    def __init__(self, n_in, hidden_dims, gaussian=False, random_order=False, seed=None):
        super().__init__()
        np.random.seed(seed)
        self.n_in = n_in
        self.n_out = 2 * n_in if gaussian else n_in
        self.hidden_dims = hidden_dims
        self.random_order = random_order
        self.gaussian = gaussian
        self.masks = {}
        self.mask_matrix = []
        dims = [n_in, *hidden_dims, self.n_out]
        layers = []
        for i in range(len(dims) - 2):
            layers.append(SimpleMaskedLinear(dims[i], dims[i + 1]))
            layers.append(nn.ReLU())
        layers.append(SimpleMaskedLinear(dims[-2], dims[-1]))
        self.model = nn.Sequential(*layers)
        self._create_masks()

    def _linears(self):
        return [m for m in self.model if isinstance(m, SimpleMaskedLinear)]

    def _create_masks(self):
        L, D = len(self.hidden_dims), self.n_in
        self.masks[0] = permutation(D) if self.random_order else np.arange(D)
        for l in range(L):
            self.masks[l + 1] = randint(low=self.masks[l].min(), high=D - 1, size=self.hidden_dims[l])
        self.masks[L + 1] = self.masks[0]
        for i in range(L + 1):
            m, m_next = self.masks[i], self.masks[i + 1]
            self.mask_matrix.append(torch.from_numpy((m_next[:, None] >= m[None, :]).astype(np.float32)))
        if self.gaussian:
            m = self.mask_matrix.pop(-1)
            self.mask_matrix.append(torch.cat((m, m), dim=0))
        self.set_masks(self.mask_matrix)

    def set_masks(self, mask_matrix):
        """Install mask matrices (e.g. a reference model's `made.mask_matrix`)."""
        lin = self._linears()
        if len(mask_matrix) != len(lin):
            raise ValueError('expected %d masks' % len(lin))
        self.mask_matrix = [torch.as_tensor(m) for m in mask_matrix]
        for layer, m in zip(lin, self.mask_matrix):
            if tuple(m.shape) != tuple(layer.weight.shape):
                raise ValueError('mask shape %s does not match weight %s' % (tuple(m.shape), tuple(layer.weight.shape)))
            layer.initialise_mask(m)

    def forward(self, x):
        lin = self._linears()
        h = x
        for i, layer in enumerate(lin):
            h = layer(h, relu_out=i < len(lin) - 1)
        return h if self.gaussian else torch.sigmoid(h)


class MaskedAutoregressiveFlow(nn.Module):
    """x = (z - mu) * exp(logp/2) with (mu, logp) = SimpleMADE(z) (reference flows.py:230-263)."""

    def __init__(self, dim, hidden_sizes, reverse_input=True, log_params={"log": False}):
        super().__init__()
        self.log_params = log_params
        self.dim = dim
        self.made = SimpleMADE(dim, hidden_sizes, gaussian=True, seed=None)
        self.reverse_input = reverse_input
        self.double()          # the reference only runs this class under a float64 default

    def load_reference(self, state_dict, mask_matrix):
        self.load_state_dict(state_dict, strict=True)
        self.made.set_masks(mask_matrix)
        return self

    def _recording(self, t):
        return torch.is_grad_enabled() and (t.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward(self, z):
        z = z.double()
        out = self.made(z)
        if self._recording(z):
            return MafFwdFn.apply(z, out, bool(self.reverse_input))
        return _lib.maf_fwd(z.contiguous(), out, self.reverse_input)

    def reverse(self, x):
        x = x.contiguous()
        z = torch.zeros_like(x)
        if self._recording(x):
            logJ = None
            for dim in range(self.dim):
                out = self.made(z)
                z, logJ = MafReverseStepFn.apply(x, out, z, dim, bool(self.reverse_input), dim == self.dim - 1)
            return z, logJ
        logJ = torch.empty(x.shape[0], dtype=x.dtype, device=x.device)
        for dim in range(self.dim):
            out = self.made(z)
            _lib.maf_reverse_step(x, out, z, dim, self.reverse_input, logJ if dim == self.dim - 1 else None)
        return z, logJ


# ---- the two data-flow loops of the reference's harness over a list of legacy layers (models/model.py:87-108) ----
def apply_flow_to_samples(layers, x, logq):
    """x through `layers` in order by layer.forward, logq - sum logJ (model.py:87-92; with x ~ prior and
    logq = prior.log_prob(x) this is apply_flow_to_prior, :78-85)."""
    for layer in layers:
        x, logJ = layer.forward(x)
        logq = logq - logJ.squeeze()
    return x, logq


def apply_reverse_to_samples(layers, x, logq, prior):
    """x back through the layers by layer.reverse; returns (z, prior.log_prob(z) + logq + sum logJ): the density the
    stack assigns to data x (model.py:103-108) - `-mean` of it is cfg-1's training loss."""
    for layer in reversed(list(layers)):
        x, logJ = layer.reverse(x)
        logq = logq + logJ.squeeze()
    return x, prior.log_prob(x) + logq
