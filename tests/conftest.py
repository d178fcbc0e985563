import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """npz fixture -> dict of numpy arrays (no pickles)."""
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_state_dict(fx, prefix='sd/'):
    """Rebuild a torch state_dict stored by gen_golden.sd_np ('/' stands for '.')."""
    sd = {}
    for k, v in fx.items():
        if k.startswith(prefix):
            sd[k[len(prefix):].replace('/', '.')] = torch.from_numpy(np.array(v))
    return sd


def rel_err(a, b, floor=1.0):
    """max |a-b| / max(floor, |b|) — the element-relative error used for the 1e-5 contract."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return ((a - b).abs() / b.abs().clamp_min(floor)).max().item()


@pytest.fixture(scope='session')
def has_gpu():
    return torch.cuda.is_available()


def pytest_terminal_summary(terminalreporter):
    """Measured gradient distances (tests/test_gpu_parity.py::assert_grad_close): the worst ratio to the bar."""
    mod = sys.modules.get('test_gpu_parity')
    rep = getattr(mod, 'GRAD_REPORT', None)
    if rep:
        worst = sorted(rep, key=lambda r: -r[1])[:5]
        terminalreporter.write_line('gradient parity (name, err / max|ref|, oracle fp32 floor): ' +
                                    '; '.join('%s %.2e %.2e' % r for r in worst))
