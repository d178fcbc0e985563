"""nf_made_rqs_fwd_noalloc - the one entry point that takes no workspace and allocates nothing (scratch = a static
device array inside the library) - against nf_made_rqs_fwd, through the C ABI.  The two share their kernels, so the
bar is bit equality of outputs, logabsdet and bin indices; two calls in a row, the second on a smaller input (the
array keeps the first call's contents), in the exact-fp32 and the split-fp16 matrix modes; a batch past the static
limit is refused with NF_ERR_SCRATCH before anything is launched.  Collected last on purpose (file name)."""
import pytest
import torch

from oracle import nf_oracle as O

pytestmark = pytest.mark.gpu


def _call_noalloc(lib, pf, h, x, y, lad, bins, K, Tb, inverse=0):
    return lib.nf_made_rqs_fwd_noalloc(
        h.data_ptr(), h.stride(0), pf.packed.data_ptr(), pf.ktab.data_ptr(), pf.bias.data_ptr(), x.data_ptr(),
        x.stride(0), y.data_ptr(), y.stride(0), lad.data_ptr(), 0, None if bins is None else bins.data_ptr(),
        x.shape[0], pf.inp, pf.features, K, 1, Tb, 1e-3, 1e-3, 1e-3, inverse, pf.dtype,
        torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize('mode', ['f32', 'f16x2'])
def test_noalloc_entry_point_equals_workspace_entry_point(mode):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import normalizing_flows_amd as nf
    from normalizing_flows_amd import _lib
    lib = _lib.load()
    D, H, K, Tb = 100, 64, 8, 3.0
    sd = O.seeded_made_state_dict(D, H, 3 * K - 1, seed=11, scale=0.1)
    w, m, b = (sd['autoregressive_net.final_layer.' + k].cuda() for k in ('weight', 'mask', 'bias'))
    dtype = {'f32': _lib.NF_DTYPE_F32, 'f16x2': _lib.NF_DTYPE_F16X2}[mode]
    pf = _lib.PackedFusedFinal(w, m, b, D, K, 'linear', dtype)
    g = torch.Generator().manual_seed(5)
    # warm-up outside any timed / captured region: B = 0 resolves this device's scratch address, launches nothing
    z = torch.empty(0, D, device='cuda')
    assert _call_noalloc(lib, pf, torch.empty(0, H, device='cuda'), z, z.clone(), torch.empty(0, device='cuda'), None,
                         K, Tb) == 0
    for B in (700, 130):                                 # the second call is smaller: stale partials sit behind it
        h = torch.randn(B, H, generator=g).cuda()
        x = (torch.randn(B, D, generator=g) * 1.4).cuda()
        for inverse in (0, 1):
            want_y, want_lad, want_bins = _lib.made_rqs(h, pf, x, Tb, inverse=bool(inverse), want_bins=True)
            y = torch.full_like(x, float('nan'))
            lad = torch.full((B,), float('nan'), device='cuda')
            bins = torch.full((B, D), -7, dtype=torch.int32, device='cuda')
            torch.cuda.synchronize()
            before = torch.cuda.memory_allocated()
            assert _call_noalloc(lib, pf, h, x, y, lad, bins, K, Tb, inverse) == 0
            torch.cuda.synchronize()
            assert torch.cuda.memory_allocated() == before
            assert torch.equal(y, want_y) and torch.equal(lad, want_lad) and torch.equal(bins, want_bins)
    # the oracle agrees with what both produce (the last batch, forward direction): 1e-5 of its fp64 result, or - for
    # the few ill-conditioned elements - 4x the oracle's own fp32-vs-fp64 distance (the rule of test_gpu_parity.py)
    pm = (w * m).cpu()
    p64 = h.cpu().double() @ pm.double().t() + b.cpu().double()
    ref_y, ref_lad = O._rq_ar_elementwise(x.cpu().double(), p64, K, Tb, False, (1e-3, 1e-3, 1e-3))
    y32, lad32 = O._rq_ar_elementwise(x.cpu(), h.cpu() @ pm.t() + b.cpu(), K, Tb, False, (1e-3, 1e-3, 1e-3))
    got_y, got_lad, _ = _lib.made_rqs(h, pf, x, Tb)
    for got, ref, ref32 in ((got_y, ref_y, y32), (got_lad, ref_lad, lad32)):
        e = (got.cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)
        floor = ((ref32.double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
        assert e.max().item() <= max(1e-5, 4 * floor), (e.max().item(), floor)
        assert (e <= 1e-5).double().mean().item() >= 0.99
    # past the static limit: refused, nothing launched (72-tile layers stop at B = 29127; here tiles = ceil(100/11))
    tiles = lib.nf_made_rqs_tiles(D, K, 1)
    too_many = (8 << 20) // (4 * tiles) + 1
    assert lib.nf_made_rqs_fwd_noalloc(h.data_ptr(), H, pf.packed.data_ptr(), pf.ktab.data_ptr(), pf.bias.data_ptr(),
                                       x.data_ptr(), D, y.data_ptr(), D, lad.data_ptr(), 0, None, too_many, H, D, K, 1,
                                       Tb, 1e-3, 1e-3, 1e-3, 0, dtype, None) == -6
    assert b'8 MiB' in lib.nf_strerror(-6)


def test_noalloc_split_mode_when_the_term_planes_do_not_fit():
    """NF_DTYPE_F16X2 at H = 512, B = 8192: the pre-split term planes of the hidden state (2 x B x 512 x 2 bytes = 16 MiB)
    do not fit the 8 MiB static array, so the call takes the kernel that splits the hidden state inside its K loop
    instead of the eight-wave plane kernel - another accumulation order.  What holds there (include/nf_hip.h): bit
    equality with nf_made_rqs_fwd given a workspace of the same 8 MiB (the path is chosen from the workspace size
    alone), and the fp32 contract against the oracle, which the plane path a full workspace selects meets too."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from normalizing_flows_amd import _lib
    lib = _lib.load()
    D, H, K, Tb, B = 40, 512, 8, 3.0, 8192
    sd = O.seeded_made_state_dict(D, H, 3 * K - 1, seed=12, scale=0.05)
    w, m, b = (sd['autoregressive_net.final_layer.' + k].cuda() for k in ('weight', 'mask', 'bias'))
    pf = _lib.PackedFusedFinal(w, m, b, D, K, 'linear', _lib.NF_DTYPE_F16X2)
    assert lib.nf_made_rqs_workspace_bytes_dtype(B, D, H, K, 1, _lib.NF_DTYPE_F16X2) > (8 << 20)     # planes do not fit
    g = torch.Generator().manual_seed(6)
    for rows in (B, 3000):                               # the second call is smaller
        h = torch.randn(rows, H, generator=g).relu().cuda()
        x = (torch.randn(rows, D, generator=g) * 1.4).cuda()
        y = torch.full_like(x, float('nan'))
        lad = torch.full((rows,), float('nan'), device='cuda')
        assert _call_noalloc(lib, pf, h, x, y, lad, None, K, Tb) == 0
        ws = torch.empty((8 << 20) // 4, dtype=torch.float32, device='cuda')
        y2, lad2 = torch.empty_like(x), torch.empty_like(lad)
        assert lib.nf_made_rqs_fwd(h.data_ptr(), H, pf.packed.data_ptr(), pf.ktab.data_ptr(), pf.bias.data_ptr(),
                                   x.data_ptr(), D, y2.data_ptr(), D, lad2.data_ptr(), 0, None, ws.data_ptr(),
                                   ws.numel() * 4, rows, H, D, K, 1, Tb, 1e-3, 1e-3, 1e-3, 0, pf.dtype,
                                   torch.cuda.current_stream().cuda_stream) == 0
        assert torch.equal(y, y2) and torch.equal(lad, lad2)
        # against the reference arithmetic (the rule of test_gpu_parity.py::assert_close): 1e-5 of the oracle's fp64
        # result on 99 % of the elements, everywhere within 4x the oracle's own fp32-vs-fp64 distance - for this path
        # and for the plane path a full workspace selects (another accumulation order: they differ from each other by
        # as much as either differs from the oracle on the ill-conditioned elements)
        y3, lad3, _ = _lib.made_rqs(h, pf, x, Tb)
        pm = (w * m).cpu()
        p64 = h.cpu().double() @ pm.double().t() + b.cpu().double()
        ref_y, ref_lad = O._rq_ar_elementwise(x.cpu().double(), p64, K, Tb, False, (1e-3, 1e-3, 1e-3))
        y32, lad32 = O._rq_ar_elementwise(x.cpu(), h.cpu() @ pm.t() + b.cpu(), K, Tb, False, (1e-3, 1e-3, 1e-3))
        for got, ref, ref32 in ((y, ref_y, y32), (lad, ref_lad, lad32), (y3, ref_y, y32), (lad3, ref_lad, lad32)):
            e = (got.cpu().double() - ref).abs() / ref.abs().clamp_min(1.0)
            floor = ((ref32.double() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
            assert e.max().item() <= max(1e-5, 4 * floor), (e.max().item(), floor)
            assert (e <= 1e-5).double().mean().item() >= 0.99
