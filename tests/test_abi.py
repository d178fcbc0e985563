"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/nf_hip.h declares (no compute calls: there is no GPU here), argument validation that happens
before any launch, and the drop-in modules' constructor / state_dict surface against the reference's
own state_dicts stored in the golden fixtures."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden, golden_state_dict
import normalizing_flows_amd as nf


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'nf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nf_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = nf._lib.load()
    names = declared_symbols()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(nf._lib.EXPORTS) == names          # the ctypes table binds exactly the header
    assert lib.nf_version() >= 100
    assert lib.nf_strerror(0) == b'ok' and b'NULL' in lib.nf_strerror(-1)


def test_argument_errors_before_launch():
    lib = nf._lib.load()
    assert lib.nf_made_packed_bytes(18032, 512, 0) == 141 * 128 * 512 * 4
    assert lib.nf_made_packed_bytes(5, 3, 0) == 128 * 64 * 4
    assert lib.nf_made_packed_bytes(5, 3, 1) == 128 * 64 * 2
    assert lib.nf_made_ktab_len(18032) == 2 * 141
    # empty batches are legal and launch nothing, even with NULL pointers
    assert lib.nf_rqs_fwd(None, 8, None, None, 8 * 23, 1.0, None, 8, None, 0, None, 0, 8, 8, 1, 3.0,
                          1e-3, 1e-3, 1e-3, 0, None) == 0
    # min_bin_width * K > 1 -> NF_ERR_BINS (reference: ValueError)
    assert lib.nf_rqs_fwd(None, 8, None, None, 8 * 23, 1.0, None, 8, None, 0, None, 4, 8, 8, 1, 3.0,
                          0.2, 1e-3, 1e-3, 0, None) == -4
    # NULL data with B > 0, K out of range, bad layout enum
    assert lib.nf_rqs_fwd(None, 8, None, None, 8 * 23, 1.0, None, 8, None, 0, None, 4, 8, 8, 1, 3.0,
                          1e-3, 1e-3, 1e-3, 0, None) == -1
    assert lib.nf_rqs_fwd(None, 8, None, None, 8 * 23, 1.0, None, 8, None, 0, None, 4, 8, 99, 1, 3.0,
                          1e-3, 1e-3, 1e-3, 0, None) == -2
    assert lib.nf_affine_fwd(None, 8, None, None, 16, 7, None, 8, None, 0, 4, 8, 0, None) == -5
    assert lib.nf_masked_linear_fwd(None, 8, None, None, None, None, 0, None, None, None, 8, 4, 8, 8, 0, 0, 0,
                                    None) == -1


def test_cpu_tensors_are_refused():
    with pytest.raises(RuntimeError):
        nf._lib.rqs(torch.zeros(2, 2), torch.zeros(2, 46), 8, 'linear', 3.0)
    t = nf.MaskedAffineAutoregressiveTransform(4, 8).eval()
    with torch.no_grad(), pytest.raises(RuntimeError):
        t(torch.zeros(3, 4))


@pytest.mark.parametrize('tag,kw', [('res', {}), ('ff', dict(use_residual_blocks=False)),
                                    ('bn', dict(use_batch_norm=True)),
                                    ('ffbn', dict(use_residual_blocks=False, use_batch_norm=True)),
                                    ('nb3', dict(num_blocks=3))])
def test_made_state_dict_surface(tag, kw):
    """Same keys, shapes, masks and degrees as the reference's MADE (strict load succeeds)."""
    ref = golden_state_dict(load_golden('made_small_' + tag))
    net = nf.MADE(features=6, hidden_features=16, output_multiplier=23, **kw)
    own = net.state_dict()
    assert list(own.keys()) == list(ref.keys())
    for k, v in ref.items():
        assert own[k].shape == v.shape and own[k].dtype == v.dtype, k
        if k.endswith('.mask') or k.endswith('.degrees'):
            assert torch.equal(own[k], v), k
    net.load_state_dict(ref, strict=True)


def test_transform_state_dict_surface():
    ref = golden_state_dict(load_golden('rqar_small'))
    comp = nf.CompositeTransform([nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=16, hidden_features=32, num_bins=8, tails='linear', tail_bound=3.0) for _ in range(3)])
    assert list(comp.state_dict().keys()) == list(ref.keys())
    comp.load_state_dict(ref, strict=True)

    ref = golden_state_dict(load_golden('rq_coupling_linear'))
    layers = []
    for j in range(4):
        mask = torch.zeros(64)
        mask[(0 if j % 2 == 0 else 1)::2] = 1
        layers.append(nf.PiecewiseRationalQuadraticCouplingTransform(
            mask, lambda i, o: nf.ResidualNet(i, o, hidden_features=32, num_blocks=2), num_bins=8, tails='linear',
            tail_bound=3.0))
    comp = nf.CompositeTransform(layers)
    assert list(comp.state_dict().keys()) == list(ref.keys())
    comp.load_state_dict(ref, strict=True)
    assert comp._transforms[0].transform_net.hidden_features == 32
    assert torch.equal(comp._transforms[1].transform_features, ref['_transforms.1.transform_features'])

    # apply_unconditional_transform=True: the identity half's PiecewiseRationalQuadraticCDF and its three parameters
    ref = golden_state_dict(load_golden('rq_coupling_uncond_linear'))
    layers = []
    for j in range(3):
        mask = torch.zeros(9)
        mask[(0 if j % 2 == 0 else 1)::2] = 1
        layers.append(nf.PiecewiseRationalQuadraticCouplingTransform(
            mask, lambda i, o: nf.ResidualNet(i, o, hidden_features=32, num_blocks=2), num_bins=6, tails='linear',
            tail_bound=3.0, apply_unconditional_transform=True))
    comp = nf.CompositeTransform(layers)
    assert list(comp.state_dict().keys()) == list(ref.keys())
    comp.load_state_dict(ref, strict=True)


def test_constructor_errors_match_reference():
    with pytest.raises(ValueError):
        nf.MADE(4, 8, random_mask=True)                                # made.py:305-306
    with pytest.raises(ValueError):
        nf.CouplingTransform(torch.zeros(2, 2), lambda i, o: None)     # transforms.py:46-47
    with pytest.raises(ValueError):
        nf.CouplingTransform(torch.zeros(0), lambda i, o: None)        # transforms.py:48-49
    with pytest.raises(ValueError):
        nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(4, 8, tails='cubic')   # transforms.py:816-817
    with pytest.raises(NotImplementedError):
        nf.MaskedFeedforwardBlock(torch.arange(1, 5), 4, context_features=3)              # made.py:179-180
    d = nf.StandardNormal([4])
    with pytest.raises(TypeError):
        d.sample(0)                                                    # distributions.py:24-25
    with pytest.raises(RuntimeError):
        d.forward()                                                    # distributions.py:9-10


def test_full_flow_checkpoint_round_trip_with_reference_keys(tmp_path):
    """A whole BaseNeuralSplineFlow (transform + base distribution) exchanges checkpoints with the reference: the
    reference's base distribution is no nn.Module, so its flow state_dict is the transform's keys under `_transform.`
    and nothing else - strict=True must hold in both directions."""
    from normalizing_flows_amd import checkpoint
    ref = golden_state_dict(load_golden('rqar_small'))
    ref_flow_sd = {'_transform.' + k: v for k, v in ref.items()}

    def make():
        comp = nf.CompositeTransform([nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
            features=16, hidden_features=32, num_bins=8, tails='linear', tail_bound=3.0) for _ in range(3)])
        return nf.BaseNeuralSplineFlow(comp, nf.StandardNormal([16]))

    flow = make()
    assert list(flow.state_dict().keys()) == list(ref_flow_sd.keys())
    path = str(tmp_path / 'reference_sd.pt')
    torch.save(ref_flow_sd, path)                                        # what `torch.save(ref_flow.state_dict())` holds
    assert checkpoint.load_checkpoint(flow, path, strict=True) == {}
    ours = str(tmp_path / 'ours.pt')
    checkpoint.save_checkpoint(flow, ours, step=3)
    again = make()
    assert checkpoint.load_checkpoint(again, ours, strict=True) == {'step': 3}
    for k, v in ref_flow_sd.items():
        assert torch.equal(again.state_dict()[k], v), k
    # the anchor still follows the module across devices / dtypes
    assert flow._distribution._anchor.device == next(flow.parameters()).device
