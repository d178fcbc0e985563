"""Development check of the one-launch hidden stack: values against the fp32-MFMA path, planes decoded, timing."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normalizing_flows_amd as nf  # noqa: E402
from normalizing_flows_amd import _lib  # noqa: E402


def decode_planes(planes, row_scale, B, H):
    """[row tile of 128][K step][term][slot = 2 row + (half ^ bit3(row))][8 halfs] -> [B, H] float64."""
    steps = H // 16
    nt = (B + 127) // 128
    p = planes[:nt * steps * 2 * 256 * 16].view(torch.float16).view(nt, steps, 2, 256, 8).double()
    v = p[:, :, 0] + p[:, :, 1]                     # [nt, steps, 256, 8]
    rows = torch.arange(128, device=planes.device)
    out = torch.empty(nt, 128, steps, 2, 8, dtype=torch.float64, device=planes.device)
    for half in (0, 1):
        slot = 2 * rows + (half ^ ((rows >> 3) & 1))
        out[:, :, :, half] = v[:, :, slot].permute(0, 2, 1, 3)
    h = out.reshape(nt * 128, H)[:B]
    bits = row_scale[:2 * nt].view(torch.int32)
    mx = bits.view(torch.float32).double()
    ex = torch.floor(torch.log2(mx.clamp_min(1e-300))) + 1            # m = f * 2^ex, f in [0.5, 1)
    e = torch.where(mx > 0, 14 - ex, torch.zeros_like(ex))
    inv = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=planes.device), -e)
    grp = torch.arange(B, device=planes.device) // 64
    return h * inv[grp][:, None]


def check(D, H, blocks, B, residual=True, seed=0, scale=1.0):
    torch.manual_seed(seed)
    net = nf.MADE(features=D, hidden_features=H, output_multiplier=2, num_blocks=blocks,
                  use_residual_blocks=residual).cuda().eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(scale)
    x = torch.randn(B, D, device='cuda')
    with torch.no_grad():
        nf.set_matmul_dtype('f32')
        ref = net.forward_hidden(x).double()
        nf.set_matmul_dtype('f16x2')
        ph = net.hidden_planes_pack()
        assert ph is not None, 'unsupported'
        planes, rs, hid = _lib.made_hidden_planes(x, ph, want_hidden=True)
        torch.cuda.synchronize()
    den = ref.abs().max().clamp_min(1.0)
    e1 = ((hid.double() - ref).abs().max() / den).item()
    dec = decode_planes(planes, rs, B, H)
    e2 = ((dec - ref).abs().max() / den).item()
    print('D=%d H=%d blocks=%d res=%s B=%d: hidden err %.2e, planes err %.2e (max|h| %.3g)' %
          (D, H, blocks, residual, B, e1, e2, ref.abs().max().item()), flush=True)
    return net, x, ph


if __name__ == '__main__':
    check(6, 32, 2, 48)
    check(20, 64, 1, 100, residual=False)
    check(64, 128, 2, 1000, scale=2.0)
    check(784, 512, 2, 333)
    net, x, ph = check(784, 512, 2, 131072)
    with torch.no_grad():
        for _ in range(3):
            _lib.made_hidden_planes(x, ph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            _lib.made_hidden_planes(x, ph)
        torch.cuda.synchronize()
        print('hidden stack, 131072 rows: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
