"""Base distributions the reference's own wrapper works with (models/distributions.py:82-144): closed-form densities,
plain torch ops on any device - checked here on the CPU against the formulas the reference writes down, evaluated
independently in float64; the GPU composition with a transform is in tests/test_gpu_parity.py."""
import math

import pytest
import torch

import normalizing_flows_amd as nf


def test_uniform_base_density_and_sampling():
    u = nf.Uniform(5, -1.0, 3.0)
    x = torch.randn(7, 5)
    lp = u.log_prob(x, context=None)                      # (reference: the support is not checked, context ignored)
    assert lp.shape == (7,) and torch.allclose(lp.double(), torch.full((7,), 5 * math.log(1 / 4.0), dtype=torch.float64),
                                               atol=1e-6)
    assert torch.equal(u.log_prob(x, context=torch.zeros(7, 2)), lp)
    u.generator = torch.Generator().manual_seed(0)
    s = u.sample_n(1000)
    assert s.shape == (1000, 5) and s.min() >= -1.0 and s.max() < 3.0 and abs(s.mean().item() - 1.0) < 0.1
    assert u.sample(3, context=torch.zeros(4, 2)).shape == (4, 3, 5)
    assert u.log_prob_plus(x, torch.ones(7), -1.0).allclose(lp - 1.0)
    with pytest.raises(ValueError):
        nf.Uniform(2, 1.0, 1.0)
    assert 'Low' in dict(u.params) and not list(u.state_dict())      # nothing enters a reference-keyed state_dict


def test_plummer_density_and_sampling():
    p = nf.PlummerModel(torch.tensor([2.0]))
    x = torch.randn(9, 3, dtype=torch.float64) * 3
    want = sum(math.log(3 / (4 * math.pi * 8.0)) - 2.5 * torch.log(x[:, j] ** 2) for j in range(3))
    assert torch.allclose(p.log_prob(x), want, rtol=1e-12)
    assert torch.allclose(p.log_prob(x.float()).double(), want, rtol=1e-5)
    p.generator = torch.Generator().manual_seed(1)
    s = nf.PlummerModel().sample_n(4000) if False else p.sample_n(4000)
    assert s.shape == (4000, 3)                                       # points, not the radius tensor (docstring)
    r = s.norm(dim=1)
    # median radius of the sampler's radial law a^2 (u^(-2/3) - 1)^(-1/2): u = 1/2
    med = 4.0 * (0.5 ** (-2 / 3) - 1) ** -0.5
    assert abs(r.median().item() / med - 1) < 0.1
    assert abs(s.mean(0)).max() < 0.5 * med                            # isotropic
    assert not list(p.state_dict())
