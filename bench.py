"""Headline benchmark: log_prob samples/sec (forward + logdet) of the RQ-spline MAF at D=784.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "cfg3"): 5 x MaskedPiecewiseRationalQuadraticAutoregressiveTransform
(features 784, hidden 512, 2 residual blocks, 8 bins, linear tails +-3) + N(0,I) base density;
one step = log_prob of one batch of 262144 synthetic N(0,I) samples already resident in HBM, per
GPU (weak scaling: the batch is sharded, config 4).  Weights: nn.Linear default init under
torch.manual_seed(0) (identical on every rank); data: torch.Generator(1234 + rank).

One JSON line on rank 0, with
  roofline     - the dominant kernel (final MADE layer, masked fp32-MFMA GEMM): algorithmic FLOP per
                 launch (2 x unmasked weights x rows) / its average launch time, measured with HIP
                 events on the launch stream inside the timed steps, against the fp32 matrix peak;
  cpu_baseline - oracle/nf_oracle.py (the reference's ATen op sequence) on this box's host cores, on a
                 bounded sample, also used to check the GPU log_prob of those rows.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, H, K, T, L = 784, 512, 8, 3.0, 5
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
# HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE doubled as the
# microarch guide prescribes for gfx950, + WRITE_SIZE), collected separately: profiles/r01_h_f32_fused64_pmc.txt.
# Keyed by (dtype, fused kernel?, rows per launch); other shapes report null.
TRAFFIC_BYTES_PER_LAUNCH = {('f32', True, 131072): 5.72e9}


def build_flow(nf):
    torch.manual_seed(0)
    layers = [nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=D, hidden_features=H, num_bins=K, tails='linear', tail_bound=T, num_blocks=2)
        for _ in range(L)]
    return nf.BaseNeuralSplineFlow(nf.CompositeTransform(layers), nf.StandardNormal([D]))


def cpu_baseline(flow_cpu_sd, x_cpu, gpu_logp, reps):
    """Times the oracle (kind 'port') on the host cores; returns (samples/s, cores, parity error)."""
    from oracle import nf_oracle as O
    layers = [O.split_state_dict({k[len('_transform.'):]: v for k, v in flow_cpu_sd.items()
                                  if k.startswith('_transform.')}, j) for j in range(L)]
    funcs = [lambda v, s=s: O.rq_autoregressive_forward(s, v, K, T) for s in layers]
    best = float('inf')
    logp = None
    with torch.no_grad():
        for _ in range(reps):
            t0 = time.perf_counter()
            logp = O.flow_log_prob(funcs, x_cpu)
            best = min(best, time.perf_counter() - t0)
    err = ((gpu_logp.double() - logp.double()).abs() / logp.double().abs().clamp_min(1.0)).max().item()
    return x_cpu.shape[0] / best, torch.get_num_threads(), err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=262144, help='samples per GPU per step')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'bf16x3', 'f16x2'], help='MADE matrix-core input type')
    ap.add_argument('--cpu-sample', type=int, default=2048)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-bf16', action='store_true', help='skip the secondary bf16-MFMA measurement')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    import normalizing_flows_amd as nf
    from normalizing_flows_amd import parallel
    nf._lib.load()
    nf.set_matmul_dtype(args.dtype)

    flow = build_flow(nf)
    sd_cpu = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(dev).eval()
    B = args.batch
    x = torch.randn(B, D, device=dev, generator=torch.Generator(device=dev).manual_seed(1234 + rank))

    def step():
        with torch.no_grad():
            logp = flow.log_prob(x)
        # the log-likelihood reduction of the sharded batch: [sum, count] in fp64, one all-reduce
        return logp, parallel.reduce_log_likelihood(logp)

    for _ in range(args.warmup):
        step()

    timer = nf._lib.KernelTimer()
    nf._lib.set_timer(timer)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logp, (ll_sum, ll_count) = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    nf._lib.set_timer(None)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    kern = timer.summary()
    if rank == 0:
        total = world * B * args.steps
        out = {
            'metric': 'log_prob samples/sec (fwd+logdet), RQ-spline MAF D=784',
            'value': total / elapsed, 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.dtype == 'f32' else 'bf16-mfma/f32-accumulate',
            'data': 'synthetic N(0,I) samples, default-init weights (seed 0)',
            'config': {'workload': 'cfg3: 5x RQ-spline MAF layer D=784 H=512 K=8 tails=linear T=3, log_prob',
                       'batch_per_gpu': B, 'global_batch': world * B, 'parallelism': 'batch-sharded x%d' % world,
                       'chunk_rows': nf.transforms.CHUNK_ROWS},
            'mean_log_prob': ll_sum / ll_count,
        }
        # dominant kernel: the final MADE layer (512 -> 18032)
        final = flow._transform._transforms[0].autoregressive_net.final_layer
        nnz = float(final.mask.sum().item())
        key = [k for k in kern if k[0] in ('made_rqs', 'masked_linear') and k[2] == final.out_features]
        if key:
            fused = key[0][0] == 'made_rqs' 
            rows = key[0][3]
            n, ms = kern[key[0]]
            flop = 2.0 * nnz * rows
            peak = PEAK_F32_MFMA_TFLOPS if args.dtype == 'f32' else PEAK_BF16_MFMA_TFLOPS
            ach = flop / (ms * 1e-3) / 1e12
            out['roofline'] = {'bound': 'mfma',
                               'kernel': ('made_rqs_kernel<%s> (final layer 512->18032 fused with the spline)' if fused
                                          else 'masked_linear_%s_kernel (final layer 512->18032)') % args.dtype,
                               'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                               'traffic': TRAFFIC_BYTES_PER_LAUNCH.get((args.dtype, fused, rows)), 'launches': n, 'avg_launch_ms': ms, 'rows_per_launch': rows,
                               'algorithmic_flop_per_launch': flop,
                               'dense_equiv_tflops': 2.0 * final.out_features * final.in_features * rows / (ms * 1e-3) / 1e12}
        rq = [k for k in kern if k[0] == 'rqs']
        if rq:
            n, ms = kern[rq[0]]
            rows = rq[0][3]
            byts = rows * (D * (4 * (3 * K - 1) + 8) + 4)
            out['spline_kernel'] = {'bound': 'hbm', 'achieved': byts / (ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                                    'unit': 'GB/s', 'frac': byts / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                    'avg_launch_ms': ms, 'algorithmic_bytes_per_launch': byts}
        out['kernel_ms'] = {'%s_%dx%d_rows%d' % (k[0], k[1], k[2], k[3]): round(v[1], 4) for k, v in kern.items()}
        if args.dtype == 'f32' and not args.no_bf16:
            # secondary figure: the same step with the MADE layers on bf16 MFMA (north_star's 2e-2 contract)
            nf.set_matmul_dtype('bf16')
            with torch.no_grad():
                lp16 = flow.log_prob(x)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                lp16, _ = step()
            torch.cuda.synchronize()
            dt16 = time.perf_counter() - t1
            nf.set_matmul_dtype('f32')
            out['bf16_mode'] = {'value': B * args.steps / dt16, 'unit': 'samples/s (this rank)',
                                'ms_per_step': 1e3 * dt16 / args.steps, 'dtype': 'bf16-mfma/f32-accumulate',
                                'max_rel_diff_vs_f32_log_prob':
                                    ((lp16 - logp).abs() / logp.abs().clamp_min(1.0)).max().item()}
        f32_tf, bf16_tf, copy_gbs = nf._lib.measure_device_peaks()
        out['measured_device_peaks'] = {'mfma_f32_tflops': f32_tf, 'mfma_bf16_tflops': bf16_tf, 'copy_gbs': copy_gbs,
                                        'note': 'back-to-back MFMA issue loop / 1 GiB float4 copy on this device; the '
                                                'roofline fractions above use the nominal peaks'}
        if not args.no_cpu:
            n = min(args.cpu_sample, B)
            v, cores, err = cpu_baseline(sd_cpu, x[:n].cpu(), logp[:n].cpu(), reps=3)
            out['cpu_baseline'] = {'value': v, 'unit': 'samples/s', 'cores': cores, 'kind': 'port',
                                   'sample': 'oracle (reference ATen op sequence, fp32, torch CPU) on the first %d '
                                             'samples of rank 0, best of 3' % n}
            out['parity_max_rel_err_vs_oracle'] = err
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
