"""Data-parallel maximum-likelihood training step of the cfg-3 model (BASELINE.json config 5).

    python benchmarks/bench_train.py [--batch 32768] [--steps 5]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 benchmarks/bench_train.py

One step = forward (-mean log_prob) + backward through the hand-written kernels + bucketed gradient
all-reduce over RCCL (when WORLD_SIZE > 1) + Adam.  Per-GPU batch is fixed (weak scaling)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402
from normalizing_flows_amd import parallel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32768)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--layers', type=int, default=5)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16x2', 'bf16x3', 'bf16'],
                    help='matrix mode of the forward and input-gradient GEMMs (weight gradients: fp32 MFMA)')
    ap.add_argument('--two-launch-forward', action='store_true',
                    help='final layer + spline of the training forward as two autograd nodes (A/B of the fused forward)')
    ap.add_argument('--fused-forward-f16x2', action='store_true',
                    help='f16x2 mode: the one-launch training forward on the t4 kernel (off by default: slower end to end)')
    args = ap.parse_args()
    from normalizing_flows_amd import transforms as _tr
    if args.two_launch_forward:
        _tr.FUSED_TRAINING_FORWARD = False
    if args.fused_forward_f16x2:
        _tr.FUSED_TRAINING_FORWARD_F16X2 = True
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    D, H, K = 784, 512, 8
    nf.set_matmul_dtype(args.dtype)
    torch.manual_seed(0)
    layers = [nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(D, H, num_bins=K, tails='linear', tail_bound=3.)
              for _ in range(args.layers)]
    flow = nf.BaseNeuralSplineFlow(nf.CompositeTransform(layers), nf.StandardNormal([D])).cuda()
    # (torch's single-launch Adam: the default multi-tensor path is 12 launches of 35-65 us per step here)
    opt = torch.optim.Adam(flow.parameters(), lr=1e-4, fused=True)
    x = torch.randn(args.batch, D, device='cuda', generator=torch.Generator(device='cuda').manual_seed(99 + rank))
    losses = []
    # gradients live in flat buckets; with several ranks their all-reduces start from per-parameter hooks while backward
    # is still running (parallel.GradientSynchronizer); with one rank this only replaces the per-step zero_grad
    sync = parallel.GradientSynchronizer(flow.parameters())

    def step():
        sync.zero_grad()
        loss = -flow.log_prob(x).mean()
        loss.backward()
        sync.finish()
        opt.step()
        return loss

    for _ in range(args.warmup):
        losses.append(step().item())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    losses.append(loss.item())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({'metric': 'training samples/sec (fwd+bwd+allreduce+Adam), RQ-spline MAF D=784',
                          'value': world * args.batch * args.steps / dt, 'unit': 'samples/s', 'n_gpus': world,
                          'ms_per_step': 1e3 * dt / args.steps, 'batch_per_gpu': args.batch, 'dtype': args.dtype,
                          'losses': losses, 'peak_mem_gb': torch.cuda.max_memory_allocated() / 2**30}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
