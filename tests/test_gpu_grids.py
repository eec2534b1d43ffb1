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


# ------------------------------------------------------------------------------------------------ correlation pyramid + lookup


def _identity_lookup(levels, coords, slots, grid, E, h, w):
    """all 196 channels of the FUSED lookup + 1x1 convolution kernel, bit for bit: the convolution is given 0/1 weights
    that copy channels 0..127, then 68..195 (one fp16 value x 1.0 accumulated in fp32 and rounded back is itself)"""
    from vipe_amd.ext import droid_net_ext
    from vipe_amd.slam.update_engine import _Packed
    out = torch.zeros(E, h, w, 196, dtype=torch.float16, device=dev())
    for c0 in (0, 68):
        wt = torch.zeros(128, 200, 1, 1)
        wt[torch.arange(128), c0 + torch.arange(128)] = 1.0
        pk = _Packed(wt.half(), torch.zeros(128), dev())
        o = torch.empty(E, h, w, 128, dtype=torch.float16, device=dev())
        droid_net_ext.corr_lookup_conv1x1(levels, coords, pk.packed, pk.bias, o, act="none", slots=slots, grid=grid)
        out[..., c0:c0 + 128] = o
    return out


@pytest.mark.parametrize("grid", [(41, 73), (35, 85), (55, 55), (9, 18), (8, 8), (12, 100), (73, 41)])
def test_pyramid_build_and_fused_lookup_on_ragged_grids(grid):
    """Row a1 + a2 on grids that are not multiples of 8 x 64 (droid_net.py:56-82, correlation_kernels.cu:22-66):
    `vipe_corr_prep` + `vipe_corr_pyramid_build_prepared` into the padded blocked store, through the pool's slots.
    Level 0 against the fp32 contraction of the same fp16 maps (one rounding to half); levels 1..3 BIT-EXACT against
    avg_pool2d's arithmetic on the kernel's own previous level with the reference's floored sizes (h >> i, w >> i); the
    FUSED blocked lookup (every one of the 196 channels, windows hanging over every border) and the channels-last lookup
    BIT-EXACT against oracle/corr.py on the device pyramid."""
    from oracle import corr as ocorr
    from vipe_amd.ext import droid_net_ext
    from vipe_amd.slam.networks import CorrBlock, CorrPool
    h, w = grid
    g = torch.Generator().manual_seed(h * 1000 + w)
    nf = 4
    fmaps = torch.randn(nf, 128, h, w, generator=g).half().to(dev())
    ii = torch.tensor([0, 1, 3, 2, 3])
    jj = torch.tensor([1, 0, 3, 0, 1])
    E = ii.shape[0]
    pool = CorrPool(capacity=2)
    pool.add_edges(fmaps, ii[:2].to(dev()), jj[:2].to(dev()), frame_range=(0, 2))
    pool.add_edges(fmaps, ii[2:].to(dev()), jj[2:].to(dev()))  # frame range read back
    assert pool.blocked and pool.pool[0].dim() == 7
    lv = pool.corr_pyramid
    assert [tuple(x.shape) for x in lv] == [(E, h, w, h >> i, w >> i) for i in range(4)]
    f1, f2 = fmaps[ii.to(dev())], fmaps[jj.to(dev())]
    ref0 = torch.matmul((f1.float() / 4).reshape(E, 128, h * w).transpose(1, 2), (f2.float() / 4).reshape(E, 128, h * w))
    d0 = (lv[0].float().reshape(E, h * w, h * w) - ref0).abs()
    assert float((d0 - ref0.abs() * 2.0 ** -10).max()) <= 2.0 ** -14, "level 0 is not the rounded fp32 contraction"
    for i in range(3):
        hn, wn = h >> (i + 1), w >> (i + 1)
        x = lv[i].float().reshape(-1, h >> i, w >> i)[:, :2 * hn, :2 * wn]
        pooled = (((x[:, 0::2, 0::2] + x[:, 0::2, 1::2]) + x[:, 1::2, 0::2]) + x[:, 1::2, 1::2]) / 4.0
        assert torch.equal(pooled.half().reshape(lv[i + 1].shape), lv[i + 1]), f"level {i + 1} pooling not bit-exact"
    # the same pyramid from gathered maps (CorrBlock's constructor: prepared operands of both map sets, no slots)
    blk = CorrBlock(f1[None], f2[None])
    for a, b_ in zip(blk.corr_pyramid, lv):
        assert torch.equal(a, b_)
    # blocked -> reference -> blocked round trip of the converters reproduces what the kernel stored where it matters
    rt = droid_net_ext.pyramid_to_reference(droid_net_ext.pyramid_to_blocked(lv, h, w), h, w)
    for a, b_ in zip(rt, lv):
        assert torch.equal(a, b_)
    # lookups
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    base = torch.from_numpy(np.stack([u, v], -1))[None].repeat(E, 1, 1, 1)
    coords = base + 5.0 * torch.randn(E, h, w, 2, generator=g)
    coords[0] = torch.rand(h, w, 2, generator=g) * torch.tensor([w + 10.0, h + 10.0]) - 5.0  # anywhere, incl. outside
    coords[1, 0, :min(w, 8)] = torch.tensor([[w - 1.0, h - 1.0], [w - 0.5, h - 0.5], [float(w), float(h)], [-1.0, -1.0],
                                             [w - 4.0, 0.0], [0.0, h - 4.0], [w + 2.5, 3.0], [-50.0, -50.0]])[:min(w, 8)]
    coords = coords.to(dev()).contiguous()
    ref = ocorr.corr_lookup([x.cpu().numpy() for x in lv], coords.cpu().numpy()[None], 3)[0]  # [E,196,h,w]
    hd = pool.lookup_deferred(coords)
    assert hd[0] == "lookup" and hd[4] == (h, w)
    got = _identity_lookup(hd[1], hd[2], hd[3], hd[4], E, h, w).permute(0, 3, 1, 2).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), "fused blocked lookup not bit-exact"
    hb = blk.lookup_deferred(coords)
    got = _identity_lookup(hb[1], hb[2], hb[3], hb[4], E, h, w).permute(0, 3, 1, 2).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))
    nhwc = pool.lookup_nhwc(coords)
    assert np.array_equal(nhwc[..., :196].permute(0, 3, 1, 2).cpu().numpy().view(np.uint16), ref.view(np.uint16))
    assert torch.count_nonzero(nhwc[..., 196:]) == 0
    # slot reuse after removals: the new edges land in freed slots of a store that held other data
    pool = pool[np.array([0, 3])]
    pool.add_edges(fmaps, torch.tensor([2, 1]).to(dev()), torch.tensor([2, 3]).to(dev()), frame_range=(1, 4))
    ii2, jj2 = torch.tensor([0, 2, 2, 1]), torch.tensor([1, 0, 2, 3])
    ref_blk = CorrBlock(fmaps[ii2.to(dev())][None], fmaps[jj2.to(dev())][None])
    for a, b_ in zip(pool.corr_pyramid, ref_blk.corr_pyramid):
        assert torch.equal(a, b_)


def test_corr_block_other_dtypes_and_channel_counts_use_the_plain_kernels():
    """CorrBlock is dtype-generic in the reference (droid_net.py:94-102): fp32 maps / channel counts other than 128 take
    `vipe_corr_volume` + `vipe_avg_pool2x2` - against torch on the host."""
    import torch.nn.functional as F
    from vipe_amd.slam.networks import CorrBlock
    g = torch.Generator().manual_seed(2)
    for (C, h, w, dt, tol) in ((128, 12, 16, torch.float32, 2e-5), (64, 9, 11, torch.float32, 2e-5), (96, 8, 8, torch.float16, 2e-3)):
        f1 = torch.randn(1, 2, C, h, w, generator=g).to(dt)
        f2 = torch.randn(1, 2, C, h, w, generator=g).to(dt)
        blk = CorrBlock(f1.to(dev()), f2.to(dev()))
        vol = torch.matmul((f1.float() / 4).reshape(2, C, h * w).transpose(1, 2), (f2.float() / 4).reshape(2, C, h * w))
        vol = vol.reshape(2 * h * w, 1, h, w)
        for i, lv in enumerate(blk.corr_pyramid):
            assert tuple(lv.shape) == (2, h, w, h >> i, w >> i) and lv.dtype == dt
            assert float((lv.float().cpu().reshape(vol.shape) - vol).abs().max()) < tol * max(1.0, float(vol.abs().max()))
            if i < 3:
                vol = F.avg_pool2d(vol, 2, stride=2)


# ------------------------------------------------------------------------------------------------ the update iteration


def test_factor_graph_update_on_the_16_9_grid():
    """BASELINE configs[1]'s real input size (1280 x 720 -> 328 x 584 -> 41 x 73 grid, system.py:46-59): ONE
    FactorGraph.update at N = 12 (E = 60), sensor-depth prior on.  (i) the fused lookup on the pool's device-built pyramid
    is bit-exact vs the oracle for every edge of a sample; (ii) the operator's outputs agree with the fp32 restatement of
    UpdateModule fed the oracle's reprojection and lookup (fp16 activations: 5e-2 px on the targets); (iii) the BA step
    equals the fp64 oracle BA fed the device's own targets / weights / damping at 1e-4 relative (north_star tolerance);
    (iv) the energy of those factors decreases; (v) every tile kernel was taken: gate hoisting and the staged gates are on."""
    import bench
    from oracle import ba as oba
    from oracle import corr as ocorr
    from oracle import geom as ogeom
    from oracle import se3 as ose3
    from oracle import update_module as oum
    from vipe_amd.ext import slam_ext
    n, H, W = 12, 328, 584
    g, buf, graph = bench.build_problem(dev(), n, H, W, 3, 0, seed=77, depth_prior=True)
    ht, wd = 41, 73
    assert (g.ht, g.wd) == (ht, wd) and graph.pgate is not None and graph.corr.blocked
    E = len(g.ii)
    graph.gate_overlap_min_edges = 16  # take the staged-gates path at this edge count too
    poses0, disps0 = buf.poses[:n].cpu().numpy().copy(), buf.disps[:n, 0].cpu().numpy().copy()
    target0 = graph.target[0].cpu().numpy().copy()
    net0, inp0 = graph.f_net.float().cpu(), graph.inp.float().cpu()
    z = torch.zeros(E, dtype=torch.long, device=dev())
    coords1, _ = slam_ext.reproject(buf.poses, buf.flattened_disps, buf.intrinsics, buf.rig, graph.ii, z, graph.jj, z, graph.ii)
    # (i)
    sel = list(range(0, E, 7))
    lv_all = graph.corr.corr_pyramid
    ref = ocorr.corr_lookup([lv[sel].cpu().numpy() for lv in lv_all], coords1[sel].cpu().numpy()[None], 3)[0]
    hd = graph.corr.lookup_deferred(coords1)
    got = _identity_lookup(hd[1], hd[2], hd[3], hd[4], E, ht, wd)[sel].permute(0, 3, 1, 2).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), "lookup on the device pyramid not bit-exact"
    # two updates: the second one consumes the gate state the first staged under its BA
    graph.update(t0=1, t1=n, itrs=3)
    torch.cuda.synchronize()
    assert graph._gate_state is not None
    tg, wg = graph.target[0].cpu().numpy(), graph.weight[0].cpu().numpy()
    damping = graph.damping[:n].cpu().numpy()
    p1, d1 = buf.poses[:n].cpu().numpy(), buf.disps[:n, 0].cpu().numpy()
    assert np.isfinite(p1).all() and np.isfinite(d1).all() and np.isfinite(tg).all()
    # (ii) oracle composition of the operator's inputs and outputs
    zz = np.zeros_like(g.ii)
    rig = ose3.se3_identity(1)
    o = ogeom.reproject(poses0, disps0, (g.intrinsics / 8.0).astype(np.float32), rig, g.ii, g.jj, zz, zz, g.ii)
    c1 = o["coords"]
    assert np.abs(c1 - coords1.cpu().numpy()).max() < 2e-3
    u, v = ogeom.pixel_grid(ht, wd, np.float32)
    motn = np.clip(np.concatenate([c1 - np.stack([u, v], -1), target0 - c1], -1).transpose(0, 3, 1, 2), -64, 64)
    corr = ocorr.corr_lookup([lv.cpu().numpy() for lv in lv_all], coords1.cpu().numpy()[None], 3)
    ix = torch.from_numpy(np.unique(g.ii, return_inverse=True)[1])
    sd = {k: w_.float() for k, w_ in graph.update_op.state_dict().items()}
    with torch.no_grad():
        net2, delta, weight, eta, _ = oum.update_forward(sd, net0, inp0, torch.from_numpy(corr).float(),
                                                         torch.from_numpy(motn).float()[None].half().float(), ix)
    assert np.abs(tg - (c1 + delta[0].numpy())).max() < 0.05
    assert np.abs(wg - weight[0].numpy()).max() < 0.02
    assert np.abs(graph.f_net.float().cpu().numpy() - net2.numpy()).max() < 0.03
    # (iii) fp64 oracle BA on the device's targets / weights / eta, from the pre-update state
    kw = dict(t0=1, t1=n, n_iters=3, pose_damping=1e-3, pose_ep=0.1)
    op, od, _, _ = oba.bundle_adjustment(poses0, disps0[:, None], g.disps_sens[:, None], g.intrinsics, rig,
                                         tg.reshape(E, -1, 2), wg.reshape(E, -1, 2), damping[:, None], g.ii, g.jj, **kw)
    assert np.abs(p1 - op).max() <= 1e-4 * max(1.0, np.abs(op).max()), np.abs(p1 - op).max()
    assert np.abs(d1 - od[:, 0]).max() <= 1e-4 * np.abs(od).max(), np.abs(d1 - od[:, 0]).max()
    # (iv)
    e0 = oba.energy(poses0, disps0[:, None], g.intrinsics, rig, tg, wg, g.ii, g.jj)
    e1 = oba.energy(p1, d1[:, None], g.intrinsics, rig, tg, wg, g.ii, g.jj)
    assert e1 < e0
    # (v) second iteration through the staged gate state == the same iteration without it (fp32 summation order)
    import copy
    state = (buf.poses.clone(), buf.disps.clone(), graph.net_n.clone(), graph.target.clone(), graph.weight.clone(),
             graph.damping.clone())
    gs = graph._gate_state
    graph.update(t0=1, t1=n, itrs=3)
    a = (buf.poses[:n].clone(), buf.disps[:n].clone(), graph.net_n.clone(), graph.target.clone())
    buf.poses.copy_(state[0]); buf.disps.copy_(state[1])
    graph.net_n = state[2]; graph.target = state[3]; graph.weight = state[4]; graph.damping.copy_(state[5])
    graph._gate_state = None
    del gs, copy
    graph.update(t0=1, t1=n, itrs=3)
    b = (buf.poses[:n], buf.disps[:n], graph.net_n, graph.target)
    assert (a[2].float() - b[2].float()).abs().max().item() < 4e-3
    assert (a[3] - b[3]).abs().max().item() < 2e-2
    assert (a[0] - b[0]).abs().max().item() < 1e-4 and (a[1] - b[1]).abs().max().item() < 1e-3


def test_update_batch_on_the_16_9_grid_uses_the_volume_path():
    """hot loop B (factor_graph.py:316-394) on a 41 x 73 grid: the backend builds its pyramids with the general-grid
    kernel and matches the reference-shaped AltCorrBlock path (volume-free, fp32) up to fp16 lookup rounding."""
    import bench
    n, H, W = 8, 328, 584
    res = []
    for alt in (False, True):
        import os
        if alt:
            os.environ["VIPE_AMD_BACKEND_ALTCORR"] = "1"
        try:
            g, buf, graph = bench.build_problem(dev(), n, H, W, 2, 0, seed=5, depth_prior=False)
            graph.update_batch(itrs=2, steps=2, optimize_intrinsics=False, optimize_rig_rotation=False)
            torch.cuda.synchronize()
            res.append((buf.poses[:n].clone(), buf.disps[:n].clone(), graph.target.clone()))
        finally:
            os.environ.pop("VIPE_AMD_BACKEND_ALTCORR", None)
    assert torch.isfinite(res[0][0]).all() and torch.isfinite(res[0][1]).all()
    assert (res[0][2] - res[1][2]).abs().max().item() < 0.25   # targets: px, fp16 volume vs fp32 volume-free correlation
    assert (res[0][0] - res[1][0]).abs().max().item() < 5e-3


# ------------------------------------------------------------------------------------------------ the driver-timed code path


def test_replayed_two_step_graph_equals_eager_steps():
    """What bench.py's timed region runs (bench.capture_two_steps): two consecutive `FactorGraph.update` calls captured
    into ONE HIP graph - the operator natively sequenced on two streams, the next iteration's hidden-state gates forked
    onto a side stream through the BA's overlap hook INSIDE the capture, plan reuse and path hints on - replayed, against
    the same number of eager steps on a twin problem from the same seeds.  Equal up to the fp32 / fp64 order of the
    atomically accumulated sums (global-context pooling, reduced system): the tolerances of
    test_update_with_gate_state_on_side_stream_matches_serial_update."""
    import bench
    n = 14  # radius-3 graph: E = 72 >= 64 edges -> gate overlap ("gated") and the operator's second stream are on
    out = {}
    for mode in ("eager", "graph"):
        g, buf, graph = bench.build_problem(dev(), n, 384, 512, 3, 0, seed=99, depth_prior=True)
        E = int(graph.ii.numel())
        assert E >= graph.gate_overlap_min_edges and E >= graph.update_op.engine(dev()).op_side_min_edges
        assert graph.gate_overlap_mode == "gated"

        def step():
            graph.update(t0=1, t1=n, itrs=3)

        for _ in range(3):
            step()
        if mode == "graph":
            cg = bench.capture_two_steps(step)   # runs 2 steps, captures 2 (not executed), replays once (2 steps)
            assert graph._gate_state is not None
            cg.replay()
            cg.replay()
        else:
            for _ in range(8):
                step()
            assert graph._gate_state is not None
        torch.cuda.synchronize()
        out[mode] = (buf.poses[:n].clone(), buf.disps[:n].clone(), graph.target.clone(), graph.weight.clone(),
                     graph.net_n.clone(), graph.damping[:n].clone())
        assert all(torch.isfinite(t.float()).all() for t in out[mode])
    a, b = out["eager"], out["graph"]
    assert (a[0] - b[0]).abs().max().item() < 1e-4, "poses"
    assert (a[1] - b[1]).abs().max().item() < 1e-4 * float(a[1].abs().max()) + 1e-4, "inverse depth"
    assert (a[4].float() - b[4].float()).abs().max().item() < 1e-2, "hidden state"
    assert (a[2] - b[2]).abs().max().item() < 5e-2, "targets (px)"
    assert (a[3] - b[3]).abs().max().item() < 2e-2, "weights"
    assert (a[5] - b[5]).abs().max().item() < 1e-3, "damping (eta)"


# ------------------------------------------------------------------------------------------------ BASELINE configs[4]: 1024 x 512


def test_config5_end_to_end_at_64x128():
    """BASELINE configs[4]'s size (1024 x 512 -> 64 x 128 grid, P = 8192, 134 MB of level-0 volume per edge; pinhole model:
    the reference's panorama camera has no projection, cameras.py:357-407): pyramid build (level 0 = rounded fp32
    contraction, pooled levels bit-exact), the fused blocked lookup bit-exact vs the oracle, one FactorGraph.update whose
    BA equals the fp64 oracle fed the device's own targets / weights / damping at 1e-4, and the encoders at 1024 x 512
    against the fp32 restatement of BasicEncoder."""
    import bench
    from oracle import ba as oba
    from oracle import corr as ocorr
    from oracle import encoder as oenc
    from oracle import se3 as ose3
    from vipe_amd.ext import slam_ext
    from vipe_amd.slam.motion_filter import DroidNet
    n, H, W = 6, 512, 1024
    g, buf, graph = bench.build_problem(dev(), n, H, W, 2, 0, seed=31, depth_prior=True)
    ht, wd = 64, 128
    E = len(g.ii)
    assert (g.ht, g.wd) == (ht, wd) and graph.corr.blocked and graph.pgate is not None
    poses0, disps0 = buf.poses[:n].cpu().numpy().copy(), buf.disps[:n, 0].cpu().numpy().copy()
    # pyramid of two edges
    sel = [0, E - 1]
    lv = [x[sel] for x in graph.corr.corr_pyramid]
    ii, jj = torch.from_numpy(g.ii[sel]).to(dev()), torch.from_numpy(g.jj[sel]).to(dev())
    f1, f2 = buf.fmaps[ii, 0], buf.fmaps[jj, 0]
    ref0 = torch.matmul((f1.float() / 4).reshape(2, 128, ht * wd).transpose(1, 2), (f2.float() / 4).reshape(2, 128, ht * wd))
    d0 = (lv[0].float().reshape(2, ht * wd, ht * wd) - ref0).abs()
    assert float((d0 - ref0.abs() * 2.0 ** -10).max()) <= 2.0 ** -14
    del ref0, d0
    for i in range(3):
        x = lv[i].float().reshape(-1, ht >> i, wd >> i)
        pooled = (((x[:, 0::2, 0::2] + x[:, 0::2, 1::2]) + x[:, 1::2, 0::2]) + x[:, 1::2, 1::2]) / 4.0
        assert torch.equal(pooled.half().reshape(lv[i + 1].shape), lv[i + 1])
    # fused lookup, bit-exact on those edges
    z = torch.zeros(E, dtype=torch.long, device=dev())
    coords1, _ = slam_ext.reproject(buf.poses, buf.flattened_disps, buf.intrinsics, buf.rig, graph.ii, z, graph.jj, z, graph.ii)
    ref = ocorr.corr_lookup([x.cpu().numpy() for x in lv], coords1[sel].cpu().numpy()[None], 3)[0]
    hd = graph.corr.lookup_deferred(coords1)
    got = _identity_lookup(hd[1], hd[2], hd[3], hd[4], E, ht, wd)[sel].permute(0, 3, 1, 2).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))
    del lv, got, ref
    # one update
    graph.update(t0=1, t1=n, itrs=3)
    torch.cuda.synchronize()
    tg, wg = graph.target[0].cpu().numpy(), graph.weight[0].cpu().numpy()
    damping = graph.damping[:n].cpu().numpy()
    p1, d1 = buf.poses[:n].cpu().numpy(), buf.disps[:n, 0].cpu().numpy()
    rig = ose3.se3_identity(1)
    kw = dict(t0=1, t1=n, n_iters=3, pose_damping=1e-3, pose_ep=0.1)
    op, od, _, _ = oba.bundle_adjustment(poses0, disps0[:, None], g.disps_sens[:, None], g.intrinsics, rig,
                                         tg.reshape(E, -1, 2), wg.reshape(E, -1, 2), damping[:, None], g.ii, g.jj, **kw)
    assert np.abs(p1 - op).max() <= 1e-4 * max(1.0, np.abs(op).max()), np.abs(p1 - op).max()
    assert np.abs(d1 - od[:, 0]).max() <= 1e-4 * np.abs(od).max(), np.abs(d1 - od[:, 0]).max()
    assert oba.energy(p1, d1[:, None], g.intrinsics, rig, tg, wg, g.ii, g.jj) < \
        oba.energy(poses0, disps0[:, None], g.intrinsics, rig, tg, wg, g.ii, g.jj)
    del graph, buf
    torch.cuda.empty_cache()
    # encoders at 1024 x 512
    torch.manual_seed(0)
    dn = DroidNet()
    img = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        f_ref = oenc.encode_features({k: v.clone().float().cpu() for k, v in dn.fnet.state_dict().items()}, img)
        n_ref, i_ref = oenc.encode_context({k: v.clone().float().cpu() for k, v in dn.cnet.state_dict().items()}, img)
    enc = dn.to(dev())
    fmap = enc.encode_features(img.to(dev()).contiguous())
    net, inp = enc.encode_context(img.to(dev()).contiguous())
    for name, got, want in (("fmap", fmap, f_ref), ("net", net, n_ref), ("inp", inp, i_ref)):
        want = want.reshape(got.shape).numpy()
        err = np.abs(got.float().cpu().numpy() - want).max()
        assert tuple(got.shape[-2:]) == (ht, wd) and err < 3e-2 * max(1.0, np.abs(want).max()), (name, err)


def test_motion_filter_and_encoders_on_the_16_9_size():
    """328 x 584 frames (1280 x 720 through the reference's resize): encoders -> 41 x 73 maps vs the fp32 restatement, and
    MotionFilter.check's score (one operator application on a CorrBlock of the general-grid kernels) vs the oracle chain."""
    from oracle import corr as ocorr
    from oracle import encoder as oenc
    from oracle import update_module as oum
    from vipe_amd.slam.motion_filter import DroidNet, MotionFilter
    torch.manual_seed(0)
    dn = DroidNet()
    sd_f = {k: v.clone() for k, v in dn.fnet.state_dict().items()}
    sd_c = {k: v.clone() for k, v in dn.cnet.state_dict().items()}
    sd_u = {k: v.clone() for k, v in dn.update.state_dict().items()}
    gen = torch.Generator().manual_seed(21)
    V, H, W = 1, 328, 584
    img0 = torch.rand(V, 3, H, W, generator=gen)
    img1 = (img0 + 0.1 * torch.rand(V, 3, H, W, generator=gen)).clamp(0, 1)
    mf = MotionFilter(dn, thresh=1e9, device=dev())
    assert mf.check(img0.to(dev()).contiguous(), None) is True
    assert mf.check(img1.to(dev()).contiguous(), None) is False
    assert tuple(mf.f_fmap.shape[-2:]) == (41, 73)
    with torch.no_grad():
        f0, f1 = oenc.encode_features(sd_f, img0), oenc.encode_features(sd_f, img1)
        net, inp = oenc.encode_context(sd_c, img0)
        ht, wd = H // 8, W // 8
        assert np.abs(mf.f_fmap.float().cpu().numpy() - f0.reshape(mf.f_fmap.shape).numpy()).max() < 3e-2 * max(1.0, float(f0.abs().max()))
        pyr = [l.numpy() for l in ocorr.corr_pyramid(f0[None], f1[None])]
        coords0 = MotionFilter.coords_grid(ht, wd)[None, None].repeat(1, V, 1, 1, 1).numpy()
        corr = torch.from_numpy(ocorr.corr_lookup(pyr, coords0, 3))
        _, delta, _ = oum.update_forward(sd_u, net[None], inp[None], corr, torch.zeros(1, V, 4, ht, wd))
        ref = float(delta.norm(dim=-1)[0].mean([1, 2]).min())
    assert abs(mf.last_score - ref) < 3e-2 * max(ref, 1e-3), (mf.last_score, ref)


# ------------------------------------------------------------------------------------------------ N > 1 control flow


def test_bench_two_ranks_on_one_card_as_a_child_process():
    """`python bench.py --gpus 2` as the driver starts it, from a fresh child process that has not touched the GPU: the
    parent spawns two ranks (torch.distributed.run), each sets its problem up, captures its step graph, joins the process
    group (gloo here: both ranks share the ONE card of this box, RCCL needs a card per rank), times its clip and enters
    the result gather.  No scaling number is read off this - it covers the N > 1 control flow (spawn -> capture ->
    process group -> barrier -> all_gather -> max over ranks) on a GPU box."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VIPE_BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--keyframes", "10", "--no-cpu-baseline", "--no-secondary", "--prof-steps", "1"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["scaling"] == "weak" and line["value"] > 0
    cfg = line["config"]
    assert cfg["result_gather"]["clips"] == 2 and cfg["result_gather"]["inside_timed_region"] and cfg["state_finite"]
    assert cfg["result_gather"]["backend"] == "gloo" and cfg["launch"].startswith("hipgraph")


def test_clips_per_gpu_workers_share_the_card():
    """`bench.py --mode clips-per-gpu` as the default bench line runs it: a parent that never touches the GPU starts K
    clip-worker processes (one `SLAMSystem.run` each: READY -> go -> one JSON line), their global-BA phases take turns on
    the file lock, the trajectories come back through files.  Two workers, short clips: covers the protocol, the lock and
    the release path on a GPU box (no throughput is read off this)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import types
    args = types.SimpleNamespace(frames=20, height=128, width=512, keep_every=1)
    out = bench.clips_per_gpu_figure(args, ks=(1, 2))
    assert set(out["by_K"]) == {"1", "2"} and out["frames"] == 20
    for k, v in out["by_K"].items():
        assert v["all_finite"] and len(v["per_clip_seconds"]) == int(k) and v["frames_per_s"] > 0
        assert all(a < b <= c for a, b, c in v["phase_ends_seconds"])
    two = out["by_K"]["2"]
    assert sum(w is not None and w >= 0 for w in two["backend_lock_wait_seconds"]) == 2  # both workers took the lock
    # a worker with the release path on (what three and more workers per card run)
    env = dict(os.environ)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    pr = subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--mode", "clip-worker", "--seed", "3", "--frames", "12",
                           "--height", "128", "--width", "512", "--release-cache", "--share-card", os.path.join(root, "gpurun_out", ".lock_test")],
                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env, cwd=root)
    try:
        assert pr.stdout.readline().strip() == "READY"
        pr.stdin.write("go\n")
        pr.stdin.flush()
        res = json.loads(pr.stdout.readline())
        assert res["finite"] and res["frames"] == 12 and res["keyframes"] == 12
        assert pr.wait(timeout=120) == 0
    finally:
        if pr.poll() is None:
            pr.kill()


def test_staged_gate_state_survives_only_while_its_scratch_is_untouched():
    """One UpdateEngine serves every FactorGraph of an UpdateModule, and a staged gate state (the hidden-state part of the
    next iteration's gates, computed under the BA) lives in that engine's scratch buffers: when ANOTHER graph runs the
    operator between two updates of the first, the staged state must be dropped, not consumed stale."""
    import bench
    from vipe_amd.slam.factor_graph import FactorGraph

    def run(interleave):
        g, buf, A = bench.build_problem(dev(), 10, 128, 512, 3, 0, seed=3, depth_prior=False)
        g2, buf2, tmp = bench.build_problem(dev(), 8, 128, 512, 2, 0, seed=4, depth_prior=False)
        B = FactorGraph(A.update_op, buf2, dev(), max_factors=-1)   # same module -> same engine, other buffers
        B.add_factors(torch.from_numpy(g2.ii), torch.from_numpy(g2.jj))
        del tmp
        for gr in (A, B):
            gr.gate_overlap_min_edges = 1
        A.update(t0=1, t1=10, itrs=2)
        assert A._gate_state is not None
        if interleave:
            B.update(t0=1, t1=8, itrs=2)     # overwrites the engine's pzr / extra / glo with B's
            assert not A.update_op.engine(dev()).gate_state_matches(A._gate_state, A.net_n, A.pgate)
        else:
            assert A.update_op.engine(dev()).gate_state_matches(A._gate_state, A.net_n, A.pgate)
        A.update(t0=1, t1=10, itrs=2)
        torch.cuda.synchronize()
        return buf.poses[:10].clone(), A.net_n.clone(), A.target.clone()

    a, b = run(False), run(True)
    assert (a[0] - b[0]).abs().max().item() < 1e-4
    assert (a[1].float() - b[1].float()).abs().max().item() < 1e-2
    assert (a[2] - b[2]).abs().max().item() < 5e-2


def test_slam_system_on_16_9_frames():
    """BASELINE configs[1]'s real input: `SLAMSystem.run` (system.py:186-316) over 584 x 328 frames (1280 x 720 through the
    reference's resize; 41 x 73 grid) - motion filter, encoders, keyframe frontend with proximity edges, the backend's
    global BA on the volume path, pass 2 through the InnerFiller, the map.  Random-init weights: pinned is the bookkeeping
    (keyframes, one unit-quaternion pose per frame, map timestamps) and that every tile kernel is the one that ran."""
    from vipe_amd.ext.lietorch import SE3
    from vipe_amd.slam.frontend import FrontendArgs
    from vipe_amd.slam.inner_filler import InfillArgs
    from vipe_amd.slam.system import Frame, SLAMConfig, SLAMSystem

    gen = torch.Generator().manual_seed(5)
    T, H, W = 20, 328, 584
    rgb = torch.rand(T, H, W, 3, generator=gen).to(dev())
    depth = (1.0 + 4.0 * torch.rand(T, H, W, generator=gen)).to(dev())
    intr = torch.tensor([525.6, 525.6, 292.0, 164.0])
    frames = []
    for t in range(T):
        pose = SE3(torch.tensor([[-0.05 * t, 0, 0, 0, 0, 0, 1.0]], device=dev())).inv()  # camera -> world
        frames.append(Frame(rgb=rgb[t], metric_depth=depth[t], intrinsics=intr, pose=SE3(pose.data[0]),
                            mask=torch.ones(H, W, dtype=torch.bool, device=dev())))
    torch.manual_seed(0)
    cfg = SLAMConfig(buffer=48, filter_thresh=0.0, frontend_backend_iters=(10,),
                     frontend=FrontendArgs(keyframe_thresh=0.0), infill=InfillArgs(infill_chunk_size=8))
    sysm = SLAMSystem(dev(), cfg)
    out = sysm.run(frames)
    torch.cuda.synchronize()
    assert out.keyframe_ids.tolist() == list(range(T))
    assert out.trajectory.data.shape == (T, 7) and bool(torch.isfinite(out.trajectory.data).all())
    assert (out.trajectory.data[:, 3:].norm(dim=-1) - 1).abs().max().item() < 1e-4
    assert torch.allclose(out.intrinsics[0].cpu(), intr)
    g = sysm.frontend.graph
    assert (g.ht, g.wd) == (41, 73) and g.corr.blocked and g.pgate is not None and g.corr.pool[0].dim() == 7
    assert tuple(sysm.buffer.fmaps.shape[-2:]) == (41, 73)
    assert out.slam_map is not None and len(out.slam_map.dense_disp_frame_inds) > 0


def test_slam_system_on_a_two_camera_rig():
    """`SLAMSystem.run` with two views per step and a rig (system.py:208-233: `n_views = len(video_streams)`, "Need rig for
    multiple views"): the multi-view branches of every stage - per-view motion-filter score (minimum over the views),
    keyframes with [V, ...] features, cross-view self edges in the frontend and the backend, per-view sensor disparities,
    pass 2's batched encoder call sliced per frame and view, the rig in the output.  Random-init weights: pinned is the
    bookkeeping."""
    from vipe_amd.ext.lietorch import SE3
    from vipe_amd.slam.frontend import FrontendArgs
    from vipe_amd.slam.inner_filler import InfillArgs
    from vipe_amd.slam.system import Frame, SLAMConfig, SLAMSystem

    gen = torch.Generator().manual_seed(8)
    T, V, H, W = 14, 2, 128, 512
    rgb = torch.rand(T, V, H, W, 3, generator=gen).to(dev())
    depth = (1.0 + 4.0 * torch.rand(T, V, H, W, generator=gen)).to(dev())
    intr = torch.tensor([460.8, 460.8, 256.0, 64.0])
    rig = SE3(torch.tensor([[0, 0, 0, 0, 0, 0, 1.0], [0.3, 0.0, 0.0, 0.0, 0.05, 0.0, 0.99875]], device=dev()))
    frames = [[Frame(rgb=rgb[t, v], metric_depth=depth[t, v], intrinsics=intr * (1.0 + 0.02 * v)) for v in range(V)]
              for t in range(T)]
    torch.manual_seed(0)
    cfg = SLAMConfig(buffer=40, filter_thresh=0.0, frontend_backend_iters=(), frontend=FrontendArgs(keyframe_thresh=0.0),
                     infill=InfillArgs(infill_chunk_size=4))
    sysm = SLAMSystem(dev(), cfg)
    out = sysm.run(frames, rig=rig)
    torch.cuda.synchronize()
    assert out.keyframe_ids.tolist() == list(range(T)) and sysm.buffer.n_views == V
    assert out.trajectory.data.shape == (T, 7) and bool(torch.isfinite(out.trajectory.data).all())
    assert (out.trajectory.data[:, 3:].norm(dim=-1) - 1).abs().max().item() < 1e-4
    assert out.intrinsics.shape == (V, 4) and torch.allclose(out.intrinsics[1].cpu(), intr * 1.02)
    assert out.get_view_trajectory(1).data.shape == (T, 7)
    g = sysm.frontend.graph
    assert g.cross_view and g.net_n.shape[0] == V * g.host_edges()["ii"].shape[0] and len(sysm.motion_filter.scores) == T - 1
    assert bool((sysm.buffer.disps_sens[:T] > 0).all())  # both views' sensor depth reached the prior


def test_tile_convolution_beyond_4_gib_of_input():
    """The halo source offsets are formed in 64 bits: a 384-channel input of more than 4 GiB (the heads' input of a
    1900-edge backend chunk at 48 x 64) still takes the tile kernel and is right at both ends of the tensor."""
    import torch.nn.functional as F
    from vipe_amd._lib import check, lib, ptr, stream_ptr
    from vipe_amd.slam.update_engine import _Packed
    B, H, W, cin_tot, cin, cout = 1900, 48, 64, 384, 256, 4
    assert B * H * W * cin_tot * 2 > 2**32
    x = torch.zeros(B, H, W, cin_tot, dtype=torch.float16, device=dev())
    g = torch.Generator().manual_seed(1)
    for b in (0, B // 2, B - 1):
        x[b] = (torch.randn(H, W, cin_tot, generator=g) * 0.5).half().to(dev())
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).half()
    pk = _Packed(w, torch.zeros(cout), dev())
    y = torch.zeros(B, H, W, 4, dtype=torch.float16, device=dev())
    check(lib().vipe_conv2d_nhwc_f16(ptr(x), ptr(pk.packed), ptr(pk.bias), None, ptr(y), B, H, W, cin, cin_tot, 0, cout, 4, 0,
                                     3, 3, 0, stream_ptr(x)), "conv")
    for b in (0, B // 2, B - 1):
        ref = F.conv2d(x[b, :, :, :cin].float().cpu().permute(2, 0, 1)[None], w.float(), None, padding=1)[0]
        assert (y[b].float().cpu().permute(2, 0, 1) - ref).abs().max().item() < 4e-3, b
    assert not y[1].any()


@pytest.mark.gpu
def test_cross_view_self_edges_outside_the_edge_window_on_a_ragged_grid():
    """A two-camera rig on a grid that needs prepared operand images (9 x 18), with `cross_view_idx` re-targeted the way
    `GraphBuffer.build_adaptive_cross_view_idx` leaves it: the cross-view self edge (i, i) of the NEWEST keyframes
    correlates with keyframe 0, far outside the window [lo, hi] of the host (ii, jj) the edges are added with.  The
    pyramid of every such edge must be the one of (frame i*V+v) x (frame 0*V+v'): the prepared range has to cover it
    (round-3 advisor finding: it covered frames [lo*V, (hi+1)*V) only and the kernel read outside the store).  Also:
    the library skips - does not read for - an edge whose frame is outside the prepared range."""
    from oracle import corr as ocorr
    from vipe_amd.ext import droid_net_ext
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.factor_graph import FactorGraph
    from vipe_amd.slam.networks import UpdateModule
    h, w, V, n = 9, 18, 2, 6
    buf = GraphBuffer(h * 8, w * 8, n_views=V, buffer_size=8, device=dev())
    buf.n_frames = n
    gen = torch.Generator().manual_seed(11)
    buf.fmaps[:n] = torch.randn(n, V, 128, h, w, generator=gen).half().to(dev())
    buf.nets[:n] = torch.randn(n, V, 128, h, w, generator=gen).tanh().half().to(dev())
    buf.inps[:n] = torch.randn(n, V, 128, h, w, generator=gen).relu().half().to(dev())
    buf.intrinsics[:] = torch.tensor([60.0, 60.0, w * 4.0, h * 4.0], device=dev())
    buf.cross_view_idx[4:6, :, 0] = 0          # keyframes 4, 5: cross-view partner = keyframe 0, other view
    torch.manual_seed(0)
    graph = FactorGraph(UpdateModule().eval(), buf, dev(), max_factors=-1, cross_view=True)
    ii = np.array([4, 5, 4, 5], dtype=np.int64)
    jj = np.array([4, 5, 5, 4], dtype=np.int64)    # host window [4, 5]; the self edges point at keyframe 0
    graph.add_factors(torch.from_numpy(ii), torch.from_numpy(jj))
    P_ = graph._edge_plan()
    pi, qi, pj, qj = (P_[k].cpu().numpy() for k in ("pi", "qi", "pj", "qj"))
    assert (pj[:4] == 0).all() and (pj[4:] >= 4).all()
    lv = graph.corr.corr_pyramid  # reference layout, edge order
    fm = buf.flattened_fmaps.cpu().numpy()
    for e in range(len(pi)):
        f1, f2 = fm[pi[e] * V + qi[e]].astype(np.float32), fm[pj[e] * V + qj[e]].astype(np.float32)
        want = (f1.reshape(128, -1).T @ f2.reshape(128, -1) / 16.0).reshape(h, w, h, w)
        got = lv[0][e].float().cpu().numpy()
        assert np.abs(got - want).max() < 2e-2 * max(1.0, np.abs(want).max()), e
    # direct library call with a frame outside the prepared range: the edge is skipped, the slot keeps its contents
    fmaps = buf.flattened_fmaps[:n * V].contiguous()
    idx1 = torch.tensor([8, 9], device=dev())
    idx2 = torch.tensor([9, 0], device=dev())   # frame 0 is outside [8, 12)
    levels = [torch.full(s, 7.0, dtype=torch.float16, device=dev())
              for s in droid_net_ext.pyramid_level_shapes(2, h, w, 4, droid_net_ext.BLOCKED)]
    droid_net_ext.corr_pyramid_build_indexed(fmaps, idx1, idx2, levels=levels, frame_range=(8, 12))
    torch.cuda.synchronize()
    ref = droid_net_ext.pyramid_to_reference(levels, h, w)
    assert bool((ref[0][1] == 7.0).all()), "edge with a frame outside the prepared store must be left untouched"
    assert not bool((ref[0][0] == 7.0).all())
