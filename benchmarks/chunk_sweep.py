"""Sweep of transforms.CHUNK_ROWS (rows per launch group) on the cfg-3 log_prob step: does a smaller group keep the hidden
stack's activations in the Infinity Cache?   python benchmarks/chunk_sweep.py [mode = f16x2]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'f16x2'
nf.set_matmul_dtype(mode)
flow = bench.build_flow(nf).cuda().eval()
x = torch.randn(262144, bench.D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1234))
for rows in (131072, 65536, 32768, 16384, 131072):
    nf.transforms.CHUNK_ROWS = rows
    with torch.no_grad():
        for _ in range(2):
            lp = flow.log_prob(x)
        tm = nf._lib.KernelTimer()
        nf._lib.set_timer(tm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            lp = flow.log_prob(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        nf._lib.set_timer(None)
    print(json.dumps({'chunk_rows': rows, 'ms_per_step': round(1e3 * dt, 3), 'samples_per_s': round(262144 / dt),
                      'kernel_ms': {'%s_%dx%d_rows%d' % k: round(v[1], 4) for k, v in tm.summary().items()},
                      'mean_logp': lp.double().mean().item()}), flush=True)
