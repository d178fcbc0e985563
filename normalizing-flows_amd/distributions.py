"""Density / sampling wrapper around a transform (reference models/distributions.py:7-196 and the
two data-flow loops of models/model.py:78-108).

    log_prob(x)  = base.log_prob(T(x)) + logabsdet            (distributions.py:154-157)
    sample(n)    = T^-1(z), z ~ base                           (distributions.py:159-171)
    forward(z)   = T.inverse(z);  reverse(x) = T(x)            (distributions.py:192-196)

The reference's own `SimpleNormal` cannot be used with its `BaseNeuralSplineFlow.log_prob/sample`
(wrong keyword / missing method); `StandardNormal` here implements the protocol
those methods call (`log_prob(x, context=None)`, `sample(n, context=None)`) with the same density.
"""
import math

import torch
from torch import nn


def _is_positive_int(x):
    return isinstance(x, int) and x > 0


def split_leading_dim(x, shape):
    """Reshapes the leading dim of x to `shape` (reference helper_functions.py:401-404)."""
    return torch.reshape(x, torch.Size(shape) + x.shape[1:])


def merge_leading_dims(x, num_dims):
    """Merges the first num_dims dims of x into one (reference helper_functions.py:407-414)."""
    if not _is_positive_int(num_dims):
        raise TypeError('Number of leading dims must be a positive integer.')
    if num_dims > x.dim():
        raise ValueError('Number of leading dims can\'t be greater than total number of dims.')
    return torch.reshape(x, torch.Size([-1]) + x.shape[num_dims:])


def repeat_rows(x, num_reps):
    """Each row of x repeated num_reps times along the leading dim (reference helper_functions.py:417-424)."""
    if not _is_positive_int(num_reps):
        raise TypeError('Number of repetitions must be a positive integer.')
    shape = x.shape
    return merge_leading_dims(x.unsqueeze(1).expand(shape[0], num_reps, *shape[1:]), num_dims=2)


class Distribution(nn.Module):
    """Base class (reference distributions.py:7-69)."""

    def forward(self, *args):
        raise RuntimeError('Forward method cannot be called for a Distribution object.')

    def log_prob(self, inputs, context=None):
        inputs = torch.as_tensor(inputs)
        if context is not None:
            context = torch.as_tensor(context)
            if inputs.shape[0] != context.shape[0]:
                raise ValueError('Number of input items must be equal to number of context items.')
        return self._log_prob(inputs, context)

    def _log_prob(self, inputs, context):
        raise NotImplementedError()

    def sample(self, num_samples, context=None, batch_size=None):
        if not _is_positive_int(num_samples):
            raise TypeError('Number of samples must be a positive integer.')
        if context is not None:
            context = torch.as_tensor(context)
        if batch_size is None:
            return self._sample(num_samples, context)
        if not _is_positive_int(batch_size):
            raise TypeError('Batch size must be a positive integer.')
        num_batches, num_leftover = divmod(num_samples, batch_size)
        samples = [self._sample(batch_size, context) for _ in range(num_batches)]
        if num_leftover > 0:
            samples.append(self._sample(num_leftover, context))
        return torch.cat(samples, dim=0)

    def _sample(self, num_samples, context):
        raise NotImplementedError()

    def sample_and_log_prob(self, num_samples, context=None):
        samples = self.sample(num_samples, context=context)
        if context is not None:
            # merge the context dimension with the sample dimension to call log_prob, then split it off again
            context = torch.as_tensor(context)
            samples = merge_leading_dims(samples, num_dims=2)
            context = repeat_rows(context, num_reps=num_samples)
            assert samples.shape[0] == context.shape[0]
        log_prob = self.log_prob(samples, context=context)
        if context is not None:
            samples = split_leading_dim(samples, shape=[-1, num_samples])
            log_prob = split_leading_dim(log_prob, shape=[-1, num_samples])
        return samples, log_prob


class StandardNormal(Distribution):
    """N(0, I) over `shape` features; density as SimpleNormal.log_prob (distributions.py:105-107)."""

    def __init__(self, shape, device=None):
        super().__init__()
        self._shape = torch.Size([shape] if isinstance(shape, int) else shape)
        # follows .to() / .cuda() like a buffer, but is NOT part of the state_dict: the reference's base distribution is
        # not an nn.Module, so a reference-keyed state_dict has no such entry (strict=True loads both ways)
        self.register_buffer('_anchor', torch.zeros(1, device=device), persistent=False)
        self.generator = None

    def _log_prob(self, inputs, context=None):
        return self.log_prob_plus(inputs, None, 1.0)

    def log_prob_plus(self, inputs, logabsdet, sign=1.0):
        """log N(inputs; 0, I) + sign * logabsdet.  On the GPU one launch (nf_std_normal_log_prob) instead of the
        pow / mul / add / sum / add chain, under autograd through autograd.StdNormalLogProbFn; elsewhere the reference's
        ops."""
        if inputs.shape[1:] != self._shape:
            raise ValueError('Expected input of shape {}, got {}'.format(self._shape, inputs.shape[1:]))
        n = inputs.shape[0]
        recording = torch.is_grad_enabled() and (inputs.requires_grad or
                                                 (logabsdet is not None and logabsdet.requires_grad))
        if inputs.is_cuda and inputs.dtype == torch.float32 and n > 0 and \
                (logabsdet is None or logabsdet.dtype == torch.float32):
            z = inputs.reshape(n, -1)
            if z.stride(1) != 1:
                z = z.contiguous()
            if recording:
                from .autograd import StdNormalLogProbFn
                return StdNormalLogProbFn.apply(z, logabsdet, sign)
            from . import _lib
            return _lib.std_normal_log_prob(z, logabsdet, sign)
        lp = (-0.5 * inputs.pow(2) - 0.5 * math.log(2 * math.pi)).reshape(n, -1).sum(dim=1)
        return lp if logabsdet is None else lp + sign * logabsdet

    def _sample(self, num_samples, context=None):
        if context is None:
            return torch.randn(num_samples, *self._shape, device=self._anchor.device, generator=self.generator)
        # one block of num_samples draws per context row: [context_size, num_samples, *shape]
        context_size = context.shape[0]
        return torch.randn(context_size * num_samples, *self._shape, device=self._anchor.device,
                           generator=self.generator).reshape(context_size, num_samples, *self._shape)

    # legacy spelling used by models/model.py:78-108
    def sample_n(self, batch_size):
        return self._sample(batch_size)


class Uniform(Distribution):
    """U(low, high)^dimensions as a base density (reference distributions.py:82-96, the one base its
    BaseNeuralSplineFlow.log_prob accepts, SURVEY 2.1-5: a `tails=None` coupling stack maps into [0, 1]^D).
    As upstream, log_prob is the constant dimensions * log(1 / (high - low)) for every row - the support is not
    checked - and `context` is accepted and ignored.  Follows .to() / .cuda(); samples come from `generator`."""

    def __init__(self, dimensions, low, high, device=None):
        super().__init__()
        if not float(high) > float(low):
            raise ValueError('Uniform needs low < high')
        self.shape = int(dimensions)
        self.low, self.high = float(low), float(high)
        self.params = [['Low', self.low], ['High', self.high]]
        self.register_buffer('_anchor', torch.zeros(1, device=device), persistent=False)
        self.generator = None

    def log_prob(self, inputs, context=None):           # (no shape requirement upstream either: any [B, ...] tensor)
        inputs = torch.as_tensor(inputs)
        return torch.full((inputs.shape[0],), -self.shape * math.log(self.high - self.low), dtype=inputs.dtype
                          if inputs.is_floating_point() else torch.get_default_dtype(), device=inputs.device)

    def log_prob_plus(self, inputs, logabsdet, sign=1.0):
        lp = self.log_prob(inputs)
        return lp if logabsdet is None else lp + sign * logabsdet

    def _sample(self, num_samples, context=None):
        n = num_samples if context is None else context.shape[0] * num_samples
        u = torch.rand(n, self.shape, device=self._anchor.device, generator=self.generator)
        x = u * (self.high - self.low) + self.low
        return x if context is None else x.reshape(context.shape[0], num_samples, self.shape)

    def sample_n(self, batch_size, context=None):
        return self._sample(batch_size)


class PlummerModel(Distribution):
    """The reference's Plummer-sphere target (distributions.py:113-144): log_prob(x) = sum over the columns of
    log(3 / (4 pi a^3)) - 2.5 log(x^2), exactly as upstream writes it (elementwise in x, not in |x|^2 + a^2).
    sample_n draws radii by the inverse CDF a^2 (u^(-2/3) - 1)^(-1/2) and uniform directions and returns the [B, 3]
    points; the reference's own sample_n builds the same points and then returns `self.plummer_radius` instead
    (distributions.py:144) - that defect is not reproduced."""

    def __init__(self, plummer_radius=None, device=None):
        super().__init__()
        a = torch.ones(1) if plummer_radius is None else torch.as_tensor(plummer_radius, dtype=torch.get_default_dtype())
        self.register_buffer('plummer_radius', a.to(device) if device is not None else a.clone(), persistent=False)
        self.params = [['Radius', a.mean().item()]]
        self.generator = None

    def log_prob(self, inputs, context=None):
        a = self.plummer_radius.to(inputs.dtype)
        return (torch.log(3.0 / (4.0 * math.pi * a ** 3)) - 2.5 * torch.log(inputs ** 2)).sum(dim=1)

    def _sample(self, num_samples, context=None):
        if context is not None:
            raise NotImplementedError('PlummerModel takes no context')
        dev, a = self.plummer_radius.device, self.plummer_radius
        u = torch.rand(3, num_samples, device=dev, generator=self.generator)
        r = a ** 2 * (u[0] ** (-2.0 / 3.0) - 1.0) ** -0.5
        theta, phi = 2.0 * math.pi * u[1], torch.acos(2.0 * u[2] - 1.0)
        return torch.stack([r * torch.sin(phi) * torch.cos(theta), r * torch.sin(phi) * torch.sin(theta),
                            r * torch.cos(phi)], dim=1)

    def sample_n(self, batch_size):
        return self._sample(batch_size)


class BaseNeuralSplineFlow(Distribution):
    """A transform plus a base distribution (reference distributions.py:147-196)."""

    def __init__(self, transform, distribution):
        super().__init__()
        self._transform = transform
        self._distribution = distribution

    def _log_prob(self, inputs, context):
        noise, logabsdet = self._transform(inputs, context=context)
        if hasattr(self._distribution, 'log_prob_plus'):
            return self._distribution.log_prob_plus(noise, logabsdet, 1.0)        # base density + logdet in one launch
        return self._distribution.log_prob(noise, context=context) + logabsdet

    def _sample(self, num_samples, context):
        # reference distributions.py:159-171: with a context, num_samples draws per context row
        noise = self._distribution.sample(num_samples, context=context)
        if context is not None:
            noise = merge_leading_dims(noise, num_dims=2)
            context = repeat_rows(context, num_reps=num_samples).contiguous()
        samples, _ = self._transform.inverse(noise, context=context)
        if context is not None:
            samples = split_leading_dim(samples, shape=[-1, num_samples])
        return samples

    def sample_and_log_prob(self, num_samples, context=None):
        if hasattr(self._distribution, 'log_prob_plus') and context is None:
            noise = self._distribution.sample(num_samples)
            samples, logabsdet = self._transform.inverse(noise, context=None)
            return samples, self._distribution.log_prob_plus(noise, logabsdet, -1.0)
        # reference distributions.py:173-186
        noise, log_prob = self._distribution.sample_and_log_prob(num_samples, context=context)
        if context is not None:
            noise = merge_leading_dims(noise, num_dims=2)
            context = repeat_rows(torch.as_tensor(context), num_reps=num_samples).contiguous()
        samples, logabsdet = self._transform.inverse(noise, context=context)
        if context is not None:
            samples = split_leading_dim(samples, shape=[-1, num_samples])
            logabsdet = split_leading_dim(logabsdet, shape=[-1, num_samples])
        return samples, log_prob - logabsdet

    def transform_to_noise(self, inputs, context=None):
        noise, _ = self._transform(inputs, context=context)
        return noise

    def forward(self, z):
        return self._transform.inverse(z)

    def reverse(self, x):
        return self._transform(x)


Flow = BaseNeuralSplineFlow
