"""Socket power while each kind of work runs back to back for a few seconds: the evidence behind "the fused f16x2
kernel is power-limited" (DESIGN.md section 4).  A side thread samples the amdgpu hwmon power sensor of card 0 every
~20 ms; the main thread loops one workload at a time:
   idle | bare MFMA loops (constant operands; pseudo-random rotating fp16 operands) | 1 GiB device copy |
   the fused final-layer + spline launch alone, f16x2 and f32 | the whole cfg-3 layer, f16x2
    python benchmarks/power_trace.py [seconds per workload = 3]  ->  one JSON line (profiles/r02_power_trace.json)"""
import glob
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402
from normalizing_flows_amd import _lib  # noqa: E402


def find(pattern):
    for p in sorted(glob.glob(pattern)):
        try:
            open(p).read()
            return p
        except OSError:
            pass
    return None


class Sampler:
    def __init__(self, path):
        self.path, self.rows, self.stop = path, [], False
        self.th = threading.Thread(target=self.run)
        self.th.start()

    def run(self):
        while not self.stop:
            try:
                self.rows.append((time.perf_counter(), float(open(self.path).read()) / 1e6))
            except (OSError, ValueError):
                pass
            time.sleep(0.02)

    def mean(self, t0, t1):
        v = [p for t, p in self.rows if t0 <= t <= t1]
        return (sum(v) / len(v), max(v), len(v)) if v else (float('nan'), float('nan'), 0)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    power = find('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average') or \
        find('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input')
    cap = find('/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap')
    if power is None:
        print(json.dumps({'error': 'no readable hwmon power sensor'}))
        return
    lib = _lib.load()
    D, H, K, T, B = 784, 512, 8, 3.0, 131072
    torch.manual_seed(0)
    layer = nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=D, hidden_features=H, num_bins=K, tails='linear', tail_bound=T, num_blocks=2).cuda().eval()
    x = torch.randn(B, D, device='cuda')
    scratch = torch.zeros(16, device='cuda')
    src = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
    dst = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    final = layer.autoregressive_net.final_layer
    nnz = float(final.mask.sum().item())

    def fused_only(mode):
        nf.set_matmul_dtype(mode)
        with torch.no_grad():
            hid = layer.autoregressive_net.forward_hidden(x)
            pf = final._packed_fused_final(D, K, 'linear')
            ws = _lib.fused_workspace(pf, B, x.device)
            out = torch.empty_like(x)
            lad = torch.empty(B, device='cuda')
        def run():
            _lib.made_rqs(hid, pf, x, T, out=out, logabsdet=lad, workspace=ws)
        return run, 2.0 * nnz * B

    def whole_layer(mode):
        nf.set_matmul_dtype(mode)
        def run():
            with torch.no_grad():
                layer(x)
        return run, None

    f16_run, f16_flop = fused_only('f16x2')
    f32_run, f32_flop = fused_only('f32')
    work = [
        ('idle', lambda: time.sleep(0.05), None),
        ('mfma_bf16_constant_operands', lambda: lib.nf_diag_mfma_loop(scratch.data_ptr(), 1024, 20000, _lib.NF_DTYPE_BF16, stream),
         1024 * 4 * 20000 * 4 * 32768.0),
        ('mfma_f16_random_rotating_operands', lambda: lib.nf_diag_mfma_loop(scratch.data_ptr(), 1024, 20000, _lib.NF_DTYPE_F16X2, stream),
         1024 * 4 * 20000 * 4 * 32768.0),
        ('mfma_f32_constant_operands', lambda: lib.nf_diag_mfma_loop(scratch.data_ptr(), 1024, 20000, _lib.NF_DTYPE_F32, stream),
         1024 * 4 * 20000 * 4 * 4096.0),
        ('copy_1GiB', lambda: lib.nf_diag_stream_copy(src.data_ptr(), dst.data_ptr(), 1 << 30, stream), None),
        ('fused_final_layer_spline_f16x2', f16_run, f16_flop),
        ('fused_final_layer_spline_f32', f32_run, f32_flop),
        ('whole_cfg3_layer_f16x2', whole_layer('f16x2')[0], None),
    ]
    smp = Sampler(power)
    out = {'sensor': power, 'power_cap_w': float(open(cap).read()) / 1e6 if cap else None, 'rows_per_launch': B,
           'seconds_per_workload': seconds, 'workloads': {}}
    for name, fn, flop in work:
        if name == 'fused_final_layer_spline_f16x2':
            nf.set_matmul_dtype('f16x2')
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            n += 5
        t1 = time.perf_counter()
        mean, peak, ns = smp.mean(t0 + 0.4 * seconds, t1)          # (the sensor averages: skip the ramp)
        row = {'power_w_mean': round(mean, 1), 'power_w_max': round(peak, 1), 'samples': ns,
               'ms_per_call': round(1e3 * (t1 - t0) / n, 4)}
        if flop:
            row['tflops'] = round(flop * n / (t1 - t0) / 1e12, 1)
        out['workloads'][name] = row
        time.sleep(0.5)
    smp.stop = True
    smp.th.join()
    nf.set_matmul_dtype('f32')
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
