"""Times the one-launch hidden stack of a cfg-3 layer (131072 rows).  NF_HIP_LIB selects a variant library."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402
from normalizing_flows_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
torch.manual_seed(0)
net = nf.MADE(features=784, hidden_features=512, output_multiplier=23, num_blocks=2).cuda().eval()
x = torch.randn(B, 784, device='cuda')
nf.set_matmul_dtype('f16x2')
with torch.no_grad():
    ph = net.hidden_planes_pack()
    for _ in range(5):
        _lib.made_hidden_planes(x, ph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        _lib.made_hidden_planes(x, ph)
    torch.cuda.synchronize()
print('%s: %.3f ms' % (os.path.basename(os.environ.get('NF_HIP_LIB', 'in-tree')), (time.perf_counter() - t0) / 50 * 1e3))
