"""GPU parity tests on the grids the reference's own pipeline produces.

`StandardResizeStreamProcessor` (vipe/slam/system.py:46-59) rescales every video to an area of 384 x 512 pixels keeping
the aspect ratio and crops to multiples of 8: 16:9 sources (both clips under assets/examples are 1280 x 720) arrive as
328 x 584, i.e. a 41 x 73 grid of P = 2993 cells - neither a multiple of 4 rows nor of 64 columns, with an odd pixel
count.  Everything here runs the tile kernels on such grids (flat tiling of the convolutions, padded blocked pyramid
store, fused lookup) through the C ABI and compares with the CPU oracle / torch fp32 exactly as tests/test_gpu_parity.py
does on the 48 x 64 grid.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ACT = {"none": 0, "relu": 1, "sigmoid": 2, "tanh": 3}
ACT_FN = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------ convolutions


@pytest.mark.parametrize("shape", [(2, 41, 73), (3, 5, 7), (1, 35, 85), (1, 9, 86), (1, 30, 100), (1, 6, 126),
                                   (1, 8, 130), (2, 55, 55), (1, 73, 41)])
def test_flat_tile_convolutions_against_torch_fp32(shape):
    """Every convolution shape of the update operator on ragged grids (flat-tile kernels up to 126 columns, two
    workgroups per CU up to 86; 130 columns: the per-tap gather kernel) vs torch fp32 conv2d of the same fp16-rounded
    operands; the canary value shows that nothing outside the image is written."""
    import torch.nn.functional as F
    from vipe_amd._lib import check, lib, ptr, stream_ptr
    from vipe_amd.slam.update_engine import _Packed
    B, H, W = shape
    torch.manual_seed(H * 131 + W)
    for (cin, cout, k, act) in [(128, 128, 3, "relu"), (448, 256, 3, "none"), (200, 128, 1, "relu"), (4, 128, 7, "relu"),
                                (128, 64, 3, "tanh"), (128, 384, 3, "relu"), (256, 4, 3, "none"), (128, 1, 3, "none"),
                                (128, 576, 1, "sigmoid"), (128, 64, 1, "none"), (128, 32, 3, "relu")]:
        x = (torch.randn(B, H, W, cin) * 0.5).half().to(dev())
        w = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).half()
        b = torch.randn(cout) * 0.1
        pk = _Packed(w, b, dev())
        ctot = cout + 8 if cout % 8 == 0 else cout + (4 - cout % 4) % 4 + 4
        y = torch.full((B, H, W, ctot), 7.0, dtype=torch.float16, device=dev())
        check(lib().vipe_conv2d_nhwc_f16(ptr(x), ptr(pk.packed), ptr(pk.bias), None, ptr(y), B, H, W, cin, cin, 0, cout,
                                         ctot, 4, k, k, ACT[act], stream_ptr(x)), "conv")
        ref = ACT_FN[act](F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float(), b, padding=k // 2))
        yc = y.float().cpu()
        err = (yc[..., 4:4 + cout].permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err < 4e-3, (shape, cin, cout, k, err)
        assert (yc[..., :4] == 7.0).all() and (yc[..., 4 + cout:] == 7.0).all(), (shape, cin, cout, k)


def test_flat_tile_two_source_input_and_initial_accumulators():
    """The GRU gates on a 41 x 73 grid: channels from two tensors, accumulators started from a hoisted partial sum
    (fp16 and the fp32 partial sums of the staged z|r gates)."""
    import torch.nn.functional as F
    from vipe_amd._lib import check, lib, ptr, stream_ptr
    from vipe_amd.slam.update_engine import _Packed
    torch.manual_seed(11)
    B, H, W = 2, 41, 73
    xa = (torch.randn(B, H, W, 128) * 0.5).half().to(dev())
    xb = (torch.randn(B, H, W, 320) * 0.5).half().to(dev())
    w = (torch.randn(256, 448, 3, 3) / (448 * 9) ** 0.5).half()
    full = F.conv2d(torch.cat([xa, xb], -1).float().cpu().permute(0, 3, 1, 2), w.float(), None, padding=1)
    # (a) unsplit two-source convolution
    pk = _Packed(w, torch.zeros(256), dev())
    y = torch.empty((B, H, W, 256), dtype=torch.float16, device=dev())
    check(lib().vipe_conv2d_fused(ptr(xa), 128, 0, ptr(xb), 320, 0, 128, ptr(pk.packed), ptr(pk.bias), None, 0, 0, ptr(y),
                                  256, 0, None, 0, 0, None, 0, 0, None, None, None, 0, 0, B, H, W, 448, 256, 3, 3, 0, 0,
                                  stream_ptr(xa)), "conv_fused")
    assert (y.float().cpu().permute(0, 3, 1, 2) - full).abs().max().item() < 4e-3
    # (b) part 1 as raw fp32 partial sums (EPI_PARTIAL) over channels [0,128), part 2 starts from them over [128,448)
    pk1 = _Packed(w[:, :128].contiguous(), torch.zeros(256), dev())
    pk2 = _Packed(w[:, 128:].contiguous(), torch.zeros(256), dev())
    part = torch.full((B, H, W, 256), 3.0, dtype=torch.float32, device=dev())
    zero = torch.zeros((B, H, W, 256), dtype=torch.float16, device=dev())
    check(lib().vipe_conv2d_fused(ptr(xa), 128, 0, None, 0, 0, 128, ptr(pk1.packed), ptr(pk1.bias), None, 0, 0, None, 256, 0,
                                  None, 0, 0, None, 0, 0, None, ptr(part), ptr(zero), 256, 0, B, H, W, 128, 256, 3, 3, 0, 6,
                                  stream_ptr(xa)), "conv_partial")
    ref1 = F.conv2d(xa.float().cpu().permute(0, 3, 1, 2), w[:, :128].float(), None, padding=1)
    assert (part.cpu().permute(0, 3, 1, 2) - ref1).abs().max().item() < 2e-3
    y2 = torch.empty((B, H, W, 256), dtype=torch.float16, device=dev())
    check(lib().vipe_conv2d_fused(ptr(xb), 320, 0, None, 0, 0, 320, ptr(pk2.packed), ptr(pk2.bias), None, 0, 0, ptr(y2), 256, 0,
                                  None, 0, 0, None, 0, 0, None, None, ptr(part), 256, 0, B, H, W, 320, 256, 3, 3, 0,
                                  0 | 0x100, stream_ptr(xb)), "conv_accinit_f32")
    assert (y2.float().cpu().permute(0, 3, 1, 2) - full).abs().max().item() < 4e-3
    # (c) fp16 initial accumulators
    ph = part.half()
    y3 = torch.empty((B, H, W, 256), dtype=torch.float16, device=dev())
    check(lib().vipe_conv2d_fused(ptr(xb), 320, 0, None, 0, 0, 320, ptr(pk2.packed), ptr(pk2.bias), None, 0, 0, ptr(y3), 256, 0,
                                  None, 0, 0, None, 0, 0, None, None, ptr(ph), 256, 0, B, H, W, 320, 256, 3, 3, 0, 0,
                                  stream_ptr(xb)), "conv_accinit_f16")
    assert (y3.float().cpu().permute(0, 3, 1, 2) - full).abs().max().item() < 6e-3


@pytest.mark.parametrize("grid", [(41, 73), (35, 85), (55, 55), (6, 9)])
def test_update_operator_on_ragged_grids_matches_torch_restatement(grid):
    """The whole flow-update operator (13 fused convolutions, gate-context hoisting, global-context kernel, GraphAgg) on
    ragged grids against oracle/update_module.py (torch fp32 restatement of UpdateModule.forward, pinned to the
    reference class by tests/test_oracle_golden.py): fp16 activations through ~8 layers -> 3e-2 absolute, as on 8 x 64
    in __graft_entry__.smoke().  Natively sequenced == issued kernel by kernel, bit for bit."""
    from oracle import update_module as oum
    from vipe_amd.slam.networks import UpdateModule
    from vipe_amd.slam.update_engine import segment_csr
    h, w = grid
    torch.manual_seed(0)
    um = UpdateModule().eval()
    eng = um.engine(dev())
    assert eng.supports_gate_split(h, w)
    E = 3
    gen = torch.Generator().manual_seed(5)
    net = torch.randn(1, E, 128, h, w, generator=gen).tanh().half()
    inp = torch.randn(1, E, 128, h, w, generator=gen).relu().half()
    corr = (torch.randn(1, E, 196, h, w, generator=gen) * 0.5).half()
    flow = (torch.randn(1, E, 4, h, w, generator=gen) * 2).half()
    ix = torch.tensor([0, 1, 1])
    net_d, delta_d, weight_d, eta_d, up_d = eng.forward(net.to(dev()), inp.to(dev()), corr.to(dev()), flow.to(dev()), ix.to(dev()))
    sd = {k: v.float() for k, v in um.state_dict().items()}
    with torch.no_grad():
        net_r, delta_r, weight_r, eta_r, up_r = oum.update_forward(sd, net.float(), inp.float(), corr.float(), flow.float(), ix)
    assert (net_d.float().cpu() - net_r).abs().max().item() < 0.03
    assert (delta_d.float().cpu() - delta_r).abs().max().item() < 0.05
    assert (weight_d.float().cpu() - weight_r).abs().max().item() < 0.02
    assert (eta_d.float().cpu() - eta_r).abs().max().item() < 5e-4
    assert (up_d.float().cpu() - up_r).abs().max().item() < 0.05
    # native sequencing vs single launches, with and without the hoisted gate context
    net_n = net[0].permute(0, 2, 3, 1).contiguous().to(dev())
    xbuf = torch.zeros(E, h, w, 320, dtype=torch.float16, device=dev())
    xbuf[..., :128] = inp[0].permute(0, 2, 3, 1).to(dev())
    corr_n = torch.zeros(E, h, w, 200, dtype=torch.float16, device=dev())
    corr_n[..., :196] = corr[0].permute(0, 2, 3, 1).to(dev())
    motn = flow[0].permute(0, 2, 3, 1).contiguous().to(dev())
    ixd = ix.to(dev())
    csr = segment_csr(ixd, 2)
    pg = eng.gate_context(xbuf)
    for pgate in (pg, None):
        outs = []
        for native in (True, False):
            xb = xbuf.clone()
            n2, dw, eta, _ = eng.forward_nhwc(net_n, xb, corr_n, motn, ix=ixd, n_src=2, csr=csr, pgate=pgate, native=native)
            outs.append((n2.clone(), dw.clone(), eta.clone(), xb))
        for a, b_ in zip(*outs):
            # the global-context sum is one float atomic per 256-pixel tile: bit-identical up to two tiles per image,
            # equal up to the fp32 order of that sum beyond
            assert torch.equal(a, b_) if h * w <= 512 else (a.float() - b_.float()).abs().max().item() < 2e-3
    # staged hidden-state gates (fp32 partial sums under the BA) == unsplit gates, up to fp32 summation order
    gs = eng.hidden_gate_state(net_n, pg, n_staged=2)
    n_a, dw_a, eta_a, _ = eng.forward_nhwc(net_n, xbuf.clone(), corr_n, motn, ix=ixd, n_src=2, csr=csr, pgate=pg, gate_state=gs)
    n_a, dw_a = n_a.clone(), dw_a.clone()
    n_b, dw_b, _, _ = eng.forward_nhwc(net_n, xbuf.clone(), corr_n, motn, ix=ixd, n_src=2, csr=csr, pgate=pg)
    assert (n_a.float() - n_b.float()).abs().max().item() < 4e-3 and (dw_a - dw_b).abs().max().item() < 2e-2
