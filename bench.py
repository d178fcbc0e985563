"""Headline benchmark: log_prob samples/sec (forward + logdet) of the RQ-spline MAF at D=784.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "cfg3"): 5 x MaskedPiecewiseRationalQuadraticAutoregressiveTransform
(features 784, hidden 512, 2 residual blocks, 8 bins, linear tails +-3) + N(0,I) base density;
one step = log_prob of one batch of 262144 synthetic N(0,I) samples already resident in HBM, per
GPU (weak scaling: the batch is sharded, config 4).  Weights: nn.Linear default init under
torch.manual_seed(0) (identical on every rank); data: torch.Generator(1234 + rank).

Matrix arithmetic (--dtype): 'f16x2' (default) carries every fp32 operand of the dominant layer as two
fp16 terms (power-of-two scaled) and forms each product from three fp16 MFMAs with fp32 accumulation -
fp32-level results (the 1e-5 parity bar, checked against the oracle in this run) on the fp16 matrix
pipe; 'bf16x3' = three bf16 terms, six MFMAs; 'f32' = v_mfma_f32_32x32x2_f32 (exact fp32 products);
'bf16' = operands rounded to bf16 (the 2e-2 contract).  The other modes are timed as secondary figures.

One JSON line on rank 0, with
  roofline     - the dominant kernel (final MADE layer fused with the spline): algorithmic FLOP per
                 launch (2 x unmasked weights x rows) / its average launch time, measured with HIP
                 events on the launch stream inside the timed steps, against the dense matrix peak of
                 the pipe it runs on (fp16/bf16 2.5 PF, fp32 157.3 TF); the split modes execute 3 (6)
                 MFMA flop per algorithmic flop, reported beside it;
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
# HBM-side bytes per launch of the dominant kernel.  NOT measured by this run: they come from separate rocprofv3 --pmc
# passes over the same command (FETCH_SIZE doubled as the microarch guide prescribes for gfx950, + WRITE_SIZE), whose
# summaries are committed under profiles/; the JSON line names the file next to the number (`traffic_source`).
# Keyed by (dtype, fused kernel?, rows per launch); other shapes report null.
TRAFFIC_BYTES_PER_LAUNCH = {('f32', True, 131072): (5.72e9, 'profiles/r01_h_f32_fused64_pmc.txt'),
                            ('f16x2', True, 131072): (5.85e9, 'profiles/r03_c_f16x2_pmc.txt')}
DTYPE_LABEL = {'f32': 'f32', 'bf16': 'bf16-mfma/f32-accumulate',
               'bf16x3': 'f32 as 3 bf16 terms (6 bf16 MFMA per product, f32 accumulate)',
               'f16x2': 'f32 as 2 fp16 terms (3 fp16 MFMA per product, f32 accumulate)'}
MFMA_PER_PRODUCT = {'f32': 1, 'bf16': 1, 'bf16x3': 6, 'f16x2': 3}


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


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def roofline_block(kern, final, mode):
    """The dominant kernel's roofline entry from the HIP-event launch times of one timed region (see module doc)."""
    nnz = float(final.mask.sum().item())
    key = [k for k in kern if k[0] in ('made_rqs', 'masked_linear') and k[2] == final.out_features]
    if not key:
        return None
    fused = key[0][0] == 'made_rqs'
    rows = key[0][3]
    n, ms = kern[key[0]]
    flop = 2.0 * nnz * rows
    peak = PEAK_F32_MFMA_TFLOPS if mode == 'f32' else PEAK_BF16_MFMA_TFLOPS
    ach = flop / (ms * 1e-3) / 1e12
    mpp = MFMA_PER_PRODUCT[mode]
    traffic = TRAFFIC_BYTES_PER_LAUNCH.get((mode, fused, rows), (None, None))
    return {'bound': 'mfma',
            'kernel': ('made_rqs kernel <%s> (final layer 512->18032 fused with the spline)' if fused
                       else 'masked_linear_%s_kernel (final layer 512->18032)') % mode,
            'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
            'mfma_flop_per_algorithmic_flop': mpp, 'executed_tflops': ach * mpp, 'executed_frac': ach * mpp / peak,
            'achieved_over_fp32_matrix_peak': ach / PEAK_F32_MFMA_TFLOPS,
            'traffic': traffic[0], 'traffic_source': traffic[1] and (traffic[1] + ' (separate rocprofv3 --pmc passes, not this run)'),
            'launches': n, 'avg_launch_ms': ms, 'rows_per_launch': rows, 'algorithmic_flop_per_launch': flop,
            'dense_equiv_tflops': 2.0 * final.out_features * final.in_features * rows / (ms * 1e-3) / 1e12}


def two_moons_leg(nf, dev, steps):
    """BASELINE.json configs[0] / north_star's two-moons figure: log-density of two-moons points under the 5-layer
    legacy Gaussian MAF (float64, hidden 64; reference flows.py:230-263 through model.py:103-108), GPU kernels vs the
    oracle on the host cores (bounded sample), checked against each other."""
    import numpy as np
    from oracle import nf_oracle as O
    np.random.seed(5)
    torch.manual_seed(5)
    layers = torch.nn.ModuleList([nf.MaskedAutoregressiveFlow(2, [64], reverse_input=(j % 2 == 0)) for j in range(5)])
    ref = [([l.weight.detach().clone() for l in f.made._linears()], [l.bias.detach().clone() for l in f.made._linears()],
            [torch.as_tensor(m).double() for m in f.made.mask_matrix], bool(f.reverse_input)) for f in layers]
    layers = layers.to(dev)
    prior = nf.StandardNormal([2]).to(dev)
    n = 1 << 20
    x = nf.two_moons(n, torch.Generator(device=dev).manual_seed(77), device=dev)
    with torch.no_grad():
        lp = nf.apply_reverse_to_samples(layers, x, 0, prior)[1]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            lp = nf.apply_reverse_to_samples(layers, x, 0, prior)[1]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        m = 1 << 16
        xc = x[:m].cpu()
        best = float('inf')
        for _ in range(3):
            t1 = time.perf_counter()
            want = O.legacy_reverse_to_samples(ref, xc, 0)[1]
            best = min(best, time.perf_counter() - t1)
    err = ((lp[:m].cpu() - want).abs() / want.abs().clamp_min(1.0)).max().item()
    return {'workload': 'cfg1: 5x MaskedAutoregressiveFlow(2, [64]) float64, log-density of two-moons points',
            'value': n / dt, 'unit': 'samples/s (this rank)', 'batch': n, 'ms_per_step': 1e3 * dt, 'dtype': 'f64',
            'cpu_baseline': {'value': m / best, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                             'sample': 'oracle on the first %d points, best of 3' % m},
            'parity_max_rel_err_vs_oracle': err, 'mean_log_prob': lp.double().mean().item()}


def launch_ranks(n, argv, script=None, stdout=None):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run, one per GPU,
    rendezvous on 127.0.0.1) and pass rank 0's JSON line through.  Called BEFORE this process has made any GPU call -
    the parent only waits - so nothing that has initialised the GPU is ever replaced or re-executed.
    Returns the children's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, min(16, (os.cpu_count() or 8) // max(1, n)))))
    return subprocess.run(cmd, env=env, stdout=stdout).returncode


def worker_stub(args):
    """--worker-stub: what a rank does around the timed region (process group, barrier, MAX over ranks, one line from
    rank 0) with no GPU work in between - tests/test_bench_launch.py drives `--gpus 2` through it on gloo."""
    import torch.distributed as dist
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
        dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    seen = torch.ones(1, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({'metric': 'stub', 'n_gpus': world, 'ranks_seen': int(seen.item()), 'gpus_flag': args.gpus,
                          'steps': args.steps, 'warmup': args.warmup, 'max_over_ranks': t.item()}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def coupling_cfg2_leg(nf, dev, steps, cpu_sample):
    """BASELINE.json configs[1]: 8 x PiecewiseRationalQuadraticCouplingTransform, D = 64, alternating masks,
    ResidualNet(hidden 32, 2 blocks) conditioners, K = 8, linear tails +-3, B = 1048576, log_prob in the library's
    default matrix mode (one launch per layer, nf_coupling_rqs_layer_fwd: fp32 MFMA hidden stack, final Linear on
    three-term bf16 MFMA - fp32-level products).  Reference path: transforms.py:80-106, :212-247, nn/resnet.py:90-100.
    HBM accounting per SURVEY 8(d): 3460 algorithmic bytes per sample per layer; spline elements = B x 32 per layer."""
    from oracle import nf_oracle as O
    Dc, HID, Kc, Lc, Bc = 64, 32, 8, 8, 1 << 20
    torch.manual_seed(0)
    layers = []
    for i in range(Lc):
        mask = torch.ones(Dc)
        mask[i % 2::2] = -1        # alternating binary mask (reference helper_functions.py:461-472)
        layers.append(nf.PiecewiseRationalQuadraticCouplingTransform(
            mask, lambda i_, o_: nf.ResidualNet(i_, o_, hidden_features=HID, num_blocks=2),
            num_bins=Kc, tails='linear', tail_bound=T))
    flow = nf.BaseNeuralSplineFlow(nf.CompositeTransform(layers), nf.StandardNormal([Dc]))
    sd = {k: v.clone() for k, v in flow.state_dict().items()}
    flow = flow.to(dev).eval()
    x = torch.randn(Bc, Dc, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))
    prev = nf.get_matmul_dtype()
    nf.set_matmul_dtype('f32')
    try:
        with torch.no_grad():
            lp = flow.log_prob(x)
            tm = nf._lib.KernelTimer()
            nf._lib.set_timer(tm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                lp = flow.log_prob(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            nf._lib.set_timer(None)
    finally:
        nf._lib.set_timer(None)
        nf.set_matmul_dtype(prev)
    kern = tm.summary()
    byts = Lc * 3460
    out = {'workload': 'cfg2: 8x RQ-spline coupling layer D=64 K=8 ResidualNet(32, 2 blocks) tails=linear T=3, log_prob',
           'value': Bc / dt, 'unit': 'samples/s (this rank)', 'batch': Bc, 'ms_per_step': 1e3 * dt, 'dtype': 'f32',
           'roofline': {'bound': 'hbm', 'achieved': Bc * byts / dt / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': Bc * byts / dt / 1e9 / PEAK_HBM_GBS, 'algorithmic_bytes_per_sample': byts,
                        'note': 'whole step (all launches of the 8 layers) against SURVEY 8(d)\'s 3460 B per sample '
                                'and layer'},
           'kernel_ms': {'%s_%dx%d_rows%d' % (k[0], k[1], k[2], k[3]): round(v[1], 4) for k, v in kern.items()}}
    fused = [k for k in kern if k[0] in ('coupling_layer', 'made_rqs')]
    if fused:
        # (coupling_layer: the whole layer in one launch - hidden stack, final Linear, spline, row copy)
        out['spline_elements_per_s_in_fused_kernel'] = fused[0][3] * (Dc // 2) / (kern[fused[0]][1] * 1e-3)
        out['layer_kernel'] = {'name': fused[0][0], 'avg_launch_ms': kern[fused[0]][1], 'launches': kern[fused[0]][0],
                               'hbm_bytes_per_launch': Bc * Dc * 8,
                               'hbm_gbs': Bc * Dc * 8 / (kern[fused[0]][1] * 1e-3) / 1e9}
    if cpu_sample:
        n = min(cpu_sample, Bc)
        xc = x[:n].cpu()
        tsd = {k[len('_transform.'):]: v for k, v in sd.items() if k.startswith('_transform.')}
        funcs = []
        for j in range(Lc):
            s_ = O.split_state_dict(tsd, j)
            funcs.append(lambda v, s_=s_: O.rq_coupling(s_, v, Kc, 'linear', T, hidden_features=HID))
        best = float('inf')
        with torch.no_grad():
            for _ in range(2):
                t1 = time.perf_counter()
                ref = O.flow_log_prob(funcs, xc)
                best = min(best, time.perf_counter() - t1)
        out['parity_max_rel_err_vs_oracle'] = ((lp[:n].cpu().double() - ref.double()).abs()
                                               / ref.double().abs().clamp_min(1.0)).max().item()
        out['cpu_baseline'] = {'value': n / best, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                               'sample': 'oracle on the first %d samples, best of 2' % n}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=262144, help='samples per GPU per step')
    ap.add_argument('--dtype', default='f16x2', choices=['f32', 'bf16', 'bf16x3', 'f16x2'],
                    help='matrix arithmetic of the dominant layer (see the module docstring)')
    ap.add_argument('--cpu-sample', type=int, default=2048)
    ap.add_argument('--stream-total', type=int, default=0,
                    help='cfg-4: this many samples per GPU per step, generated on the device and pushed through in '
                         'groups of --batch rows (nothing of size [total, D] is ever resident)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-bf16', action='store_true', help='skip the secondary measurements in the other modes')
    ap.add_argument('--worker-stub', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit('bench.py: --gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # no launcher around us: become one (fresh child processes; this parent never touches the GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or run `python bench.py '
                 '--gpus %d` without a launcher, which starts the ranks itself)' % (args.gpus, world, args.gpus, args.gpus))
    if args.worker_stub:
        return worker_stub(args)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        ranks_seen = dist.get_world_size()
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

    gen = torch.Generator(device=dev).manual_seed(4321 + rank)
    xbuf = torch.empty_like(x) if args.stream_total else None

    last = {}

    def make_rows(s, e):                     # cfg-4: the shard is produced pass by pass on the device
        return xbuf[:e - s].normal_(generator=gen)

    def log_prob_pass(xb):
        last['logp'] = flow.log_prob(xb)
        return last['logp']

    def step():
        if args.stream_total:
            # cfg-4 (parallel.streamed_log_likelihood): only an fp64 [sum, count] pair survives a pass; one all-reduce
            with torch.no_grad():
                s_, n_, _ = parallel.streamed_log_likelihood(log_prob_pass, world * args.stream_total, make_rows, B)
            return last['logp'], (s_, n_)
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
        per_step = args.stream_total if args.stream_total else B
        total = world * per_step * args.steps
        out = {
            'metric': 'log_prob samples/sec (fwd+logdet), RQ-spline MAF D=784',
            'value': total / elapsed, 'unit': 'samples/s', 'n_gpus': world, 'ranks_seen': ranks_seen, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': DTYPE_LABEL[args.dtype],
            'data': 'synthetic N(0,I) samples, default-init weights (seed 0)',
            'config': {'workload': 'cfg3: 5x RQ-spline MAF layer D=784 H=512 K=8 tails=linear T=3, log_prob',
                       'batch_per_gpu': per_step, 'global_batch': world * per_step, 'rows_per_pass': B,
                       'streamed': bool(args.stream_total), 'parallelism': 'batch-sharded x%d' % world,
                       'chunk_rows': nf.transforms.CHUNK_ROWS},
            'mean_log_prob': ll_sum / ll_count,
        }
        # dominant kernel: the final MADE layer (512 -> 18032)
        final = flow._transform._transforms[0].autoregressive_net.final_layer
        rb = roofline_block(kern, final, args.dtype)
        if rb:
            out['roofline'] = rb
        rq = [k for k in kern if k[0] == 'rqs']
        if rq:
            n, ms = kern[rq[0]]
            rows = rq[0][3]
            byts = rows * (D * (4 * (3 * K - 1) + 8) + 4)
            out['spline_kernel'] = {'bound': 'hbm', 'achieved': byts / (ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                                    'unit': 'GB/s', 'frac': byts / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                    'avg_launch_ms': ms, 'algorithmic_bytes_per_launch': byts}
        out['kernel_ms'] = {'%s_%dx%d_rows%d' % (k[0], k[1], k[2], k[3]): round(v[1], 4) for k, v in kern.items()}
        if not args.no_bf16 and not args.stream_total and world == 1:       # (secondary legs: single-GPU runs only)
            # secondary figures: the same step in the other matrix modes (this rank's shard)
            out['other_modes'] = {}
            for mode in ('f32', 'bf16x3', 'f16x2', 'bf16'):
                if mode == args.dtype:
                    continue
                nf.set_matmul_dtype(mode)
                with torch.no_grad():
                    lpm = flow.log_prob(x)
                tm = nf._lib.KernelTimer()
                nf._lib.set_timer(tm)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    lpm, _ = step()
                torch.cuda.synchronize()
                dtm = time.perf_counter() - t1
                nf._lib.set_timer(None)
                out['other_modes'][mode] = {'value': B * args.steps / dtm, 'unit': 'samples/s (this rank)',
                                            'ms_per_step': 1e3 * dtm / args.steps, 'dtype': DTYPE_LABEL[mode],
                                            'max_rel_diff_vs_headline_log_prob':
                                                ((lpm - logp).abs() / logp.abs().clamp_min(1.0)).max().item(),
                                            'roofline': roofline_block(tm.summary(), final, mode)}
                if mode == 'f32':
                    # the reference's own arithmetic (every product an exact fp32 product): its roofline entry in full
                    out['roofline_f32_mode'] = out['other_modes'][mode]['roofline']
            nf.set_matmul_dtype(args.dtype)
            # the other direction of the same flow: sample = transform.inverse(z) (reference distributions.py:192-193,
            # transforms.py:493-500: D conditioner passes per layer; here one row-resident launch per layer)
            ns = min(B, 65536)
            zs = torch.randn(ns, D, device=dev, generator=torch.Generator(device=dev).manual_seed(99))
            with torch.no_grad():
                xs, lds = flow.forward(zs)
                ti = nf._lib.KernelTimer()
                nf._lib.set_timer(ti)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                xs, lds = flow.forward(zs)
                torch.cuda.synchronize()
                dts = time.perf_counter() - t1
                nf._lib.set_timer(None)
                zb, ldb = flow.reverse(xs)
            out['inverse_sampling'] = {'value': ns / dts, 'unit': 'samples/s (this rank)', 'batch': ns,
                                       'roundtrip_max_abs_err': (zb - zs).abs().max().item(),
                                       'logdet_cancel_max_abs': (lds + ldb).abs().max().item()}
            ki = ti.summary()
            rows_k = [k for k in ki if k[0] == 'made_rqs_inverse_rows']
            if rows_k:
                # the row-resident kernel does one conditioner pass of work per sample (13.74 MFLOP unmasked MACs x 2,
                # SURVEY 8d) on fp32 MFMA; its launch time from HIP events on the launch stream
                made = flow._transform._transforms[0].autoregressive_net
                nnz_all = float(sum(l.mask.sum().item() for l in made.modules() if hasattr(l, 'mask')))
                n_l, ms_l = ki[rows_k[0]]
                ach = 2.0 * nnz_all * rows_k[0][3] / (ms_l * 1e-3) / 1e12
                out['inverse_sampling']['roofline'] = {
                    'bound': 'mfma', 'kernel': 'made_rqs_inverse_kernel (features 0..H of one layer, row-resident)',
                    'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS,
                    'launches': n_l, 'avg_launch_ms': ms_l, 'rows_per_launch': rows_k[0][3],
                    'algorithmic_flop_per_launch': 2.0 * nnz_all * rows_k[0][3], 'traffic': None}
                out['inverse_sampling']['kernel_ms'] = {'%s_%dx%d_rows%d' % (k[0], k[1], k[2], k[3]): round(v[1], 4)
                                                        for k, v in ki.items()}
            # BASELINE.json configs[1] in the same driver-timed line
            out['coupling_cfg2'] = coupling_cfg2_leg(nf, dev, max(args.steps // 2, 3), 0 if args.no_cpu else 65536)
        f32_tf, bf16_tf, copy_gbs, f16_rand_tf = nf._lib.measure_device_peaks()
        out['measured_device_peaks'] = {'mfma_f32_tflops': f32_tf, 'mfma_bf16_tflops': bf16_tf, 'copy_gbs': copy_gbs,
                                        'mfma_f16_random_operand_tflops': f16_rand_tf,
                                        'note': 'back-to-back MFMA issue loops (fp32 / bf16 on constant operands; fp16 on '
                                                'pseudo-random operands that change per instruction, 1 s sustained: what '
                                                'the pipe holds under the board power cap on real data - '
                                                'profiles/r02_power_trace.json) / 1 GiB float4 copy on this device; the '
                                                'roofline fractions above use the nominal peaks'}
        if 'roofline' in out and args.dtype != 'f32' and f16_rand_tf > 0:
            out['roofline']['executed_frac_of_measured_random_operand_peak'] = out['roofline']['executed_tflops'] / f16_rand_tf
        if not args.stream_total:
            out['two_moons'] = two_moons_leg(nf, dev, max(args.steps, 3))     # (rank 0's GPU, after the timed region)
        if not args.no_cpu and world == 1:
            n = min(args.cpu_sample, B)
            v, cores, err = cpu_baseline(sd_cpu, x[:n].cpu(), logp[:n].cpu(), reps=3)
            out['cpu_baseline'] = {'value': v, 'unit': 'samples/s', 'cores': cores, 'kind': 'port',
                                   'cpu_model': cpu_model(),
                                   'sample': 'oracle (reference ATen op sequence, fp32, torch CPU) on the first %d '
                                             'samples of rank 0, best of 3' % n}
            out['parity_max_rel_err_vs_oracle'] = err
        elif world > 1:
            out['cpu_baseline_note'] = ('the oracle is timed on rank 0 of the N=1 run only (the N=1 line carries '
                                        'cpu_baseline and the parity figure); secondary legs likewise')
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
