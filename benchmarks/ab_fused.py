"""A/B of the dominant kernel alone: the fused final-layer + spline launch of one cfg-3 layer (131072 rows, f16x2 unless
told otherwise), back to back for a few seconds (the power controller needs that long), with the board power and the
shader clock sampled beside it.  The library is the one NF_HIP_LIB names (benchmarks/build_fused_variant.sh).
    NF_HIP_LIB=... python benchmarks/ab_fused.py [seconds = 2.5] [mode = f16x2] [rows = 131072]  ->  one JSON line"""
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
    def __init__(self, power, freq):
        self.power, self.freq, self.rows, self.stop = power, freq, [], False
        self.th = threading.Thread(target=self.run)
        self.th.start()

    def run(self):
        while not self.stop:
            try:
                p = float(open(self.power).read()) / 1e6 if self.power else float('nan')
                f = float(open(self.freq).read()) / 1e6 if self.freq else float('nan')
                self.rows.append((time.perf_counter(), p, f))
            except (OSError, ValueError):
                pass
            time.sleep(0.02)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    mode = sys.argv[2] if len(sys.argv) > 2 else 'f16x2'
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
    power = find('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average') or \
        find('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input')
    freq = find('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input')
    D, H, K, T = 784, 512, 8, 3.0
    torch.manual_seed(0)
    layer = nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
        features=D, hidden_features=H, num_bins=K, tails='linear', tail_bound=T, num_blocks=2).cuda().eval()
    x = torch.randn(B, D, device='cuda')
    final = layer.autoregressive_net.final_layer
    nnz = float(final.mask.sum().item())
    nf.set_matmul_dtype(mode)
    with torch.no_grad():
        hid = layer.autoregressive_net.forward_hidden(x)
        pf = final._packed_fused_final(D, K, 'linear')
        ws = _lib.fused_workspace(pf, B, x.device)
        out = torch.empty_like(x)
        lad = torch.empty(B, device='cuda')

    def run():
        _lib.made_rqs(hid, pf, x, T, out=out, logabsdet=lad, workspace=ws)

    run()
    torch.cuda.synchronize()
    smp = Sampler(power, freq)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    n, evs = 0, []
    while time.perf_counter() - t0 < seconds:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        b.record()
        torch.cuda.synchronize()
        evs.append(a.elapsed_time(b) / 10)
        n += 10
    t1 = time.perf_counter()
    smp.stop = True
    smp.th.join()
    rows = [r for r in smp.rows if t0 + 0.4 * seconds <= r[0] <= t1]
    late = evs[len(evs) // 2:]
    print(json.dumps({'lib': os.path.basename(os.environ.get('NF_HIP_LIB', 'in-tree')), 'mode': mode, 'rows': B,
                      'ms_per_launch_settled': round(sum(late) / len(late), 4), 'ms_first_batch': round(evs[0], 4),
                      'launches': n, 'tflops_algorithmic': round(2.0 * nnz * B / (sum(late) / len(late) * 1e-3) / 1e12, 1),
                      'power_w': round(sum(r[1] for r in rows) / max(1, len(rows)), 1),
                      'sclk_mhz': round(sum(r[2] for r in rows) / max(1, len(rows)), 1),
                      'checksum': float(lad.double().sum().item())}), flush=True)


if __name__ == '__main__':
    main()
