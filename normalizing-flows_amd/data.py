"""Synthetic inputs of the benchmark configurations (BASELINE.json configs).

The reference has no data generators beyond its priors / targets (models/distributions.py:82-144); the two-moons set
of config 1 is this build's definition, fixed here so that fixtures, tests and bench.py draw the same points.
"""
import math

import torch

# analytic mean / standard deviation of the un-standardised two-moons mixture below (noise 0.1 included)
_MOONS_MEAN = (0.5, 0.25)
_MOONS_STD = (math.sqrt(1.01 - 0.25), math.sqrt(0.25 + 0.5 * (0.75 - 2.0 / math.pi) + 0.01 - 0.0625))


def two_moons(n, generator=None, device=None, dtype=torch.float64, noise=0.1):
    """n points of the two-moons set, standardised with the mixture's analytic moments: theta ~ U(0, pi); the
    upper moon is (cos t, sin t), the lower one (1 - cos t, 0.5 - sin t); + N(0, noise^2) per coordinate.  Points
    alternate upper / lower by index parity, so any even-sized batch is balanced.  Draws come from `generator`
    (on `device`): the same seed gives the same points on every machine for a CPU generator."""
    theta = torch.rand(n, generator=generator, device=device, dtype=dtype) * math.pi
    eps = torch.randn(n, 2, generator=generator, device=device, dtype=dtype) * noise
    lower = (torch.arange(n, device=device) % 2 == 1)
    c, s = torch.cos(theta), torch.sin(theta)
    x = torch.where(lower, 1.0 - c, c)
    y = torch.where(lower, 0.5 - s, s)
    pts = torch.stack([x, y], dim=1) + eps
    mean = torch.tensor(_MOONS_MEAN, device=device, dtype=dtype)
    std = torch.tensor(_MOONS_STD, device=device, dtype=dtype)
    return (pts - mean) / std
