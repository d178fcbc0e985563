import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normalizing_flows_amd as nf
from oracle import nf_oracle as O
torch.manual_seed(9)
D, H, K = 40, 64, 8
x = torch.randn(200, D, device='cuda')
for up in (1.0, 1e5, 1e-5):
    t = nf.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(D, H, num_bins=K, tails='linear', tail_bound=3.).cuda().eval()
    with torch.no_grad():
        net = t.autoregressive_net
        net.initial_layer.weight.mul_(up); net.initial_layer.bias.mul_(up)
        for blk in net.blocks:
            blk.linear_layers[1].weight.mul_(up); blk.linear_layers[1].bias.mul_(up)
            blk.linear_layers[0].weight.mul_(1.0 / up)
        net.final_layer.weight.mul_(1.0 / up)
        sd = {k: v.detach().cpu() for k, v in t.state_dict().items()}
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        y64, ld64 = O.rq_autoregressive_forward(sd64, x.cpu().double(), K, 3.0)
        y32, ld32 = O.rq_autoregressive_forward(sd, x.cpu(), K, 3.0)
        print('up', up, 'oracle fp32: y %.2e ld %.2e' % ((y32.double() - y64).abs().max().item(), (ld32.double() - ld64).abs().max().item()))
        for mode in ('f32', 'bf16x3', 'f16x2'):
            nf.set_matmul_dtype(mode)
            y, ld = t(x)
            print('   ', mode, 'y %.2e ld %.2e' % ((y.cpu().double() - y64).abs().max().item(), (ld.cpu().double() - ld64).abs().max().item()))
        nf.set_matmul_dtype('f32')
