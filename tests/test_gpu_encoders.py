"""GPU parity of the frame encoders (csrc/encoder.hip through the C ABI `vipe_enc_*`).

Layer kernels against torch fp32 convolutions of the same fp16 operands (4e-3 of the output scale), the whole
fnet / cnet against the fixture produced by the reference's own BasicEncoder classes (tests/golden/encoder_reference.npz;
fp16 activations vs the reference's fp32 CPU run: 3e-2 of the output scale, the bound used for the update operator)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vipe_amd._lib import check, lib, ptr, stream_ptr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _stats_of(y_nhwc):
    """[B,H,W,C] fp16 -> [B,C,2] fp32 (sum, sum of squares)"""
    v = y_nhwc.float().flatten(1, 2)
    return torch.stack([v.sum(1), (v * v).sum(1)], -1).contiguous()


def _inorm_relu(x_nhwc):
    v = x_nhwc.float().permute(0, 3, 1, 2)
    return F.relu(F.instance_norm(v, eps=1e-5)).half()  # [B,C,H,W] fp16, as the consumer forms it on load


@pytest.mark.parametrize("k,stride,cin,cout", [(3, 1, 32, 32), (3, 2, 32, 64), (3, 1, 64, 64), (1, 2, 32, 64),
                                                (3, 2, 64, 128), (3, 1, 128, 128), (1, 2, 64, 128), (1, 1, 128, 256)])
@pytest.mark.parametrize("mode", ["plain", "norm_on_load", "residual"])
def test_encoder_conv_against_torch_fp32(k, stride, cin, cout, mode):
    from vipe_amd.slam.encoders import _pack_conv
    torch.manual_seed(k * 100 + stride * 10 + cin)
    B, H, W = 2, 22, 42  # partial tiles in both directions
    conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2).to(dev())
    x = torch.randn(B, H, W, cin, device=dev()).half()
    wb = _pack_conv(conv)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xin = x.permute(0, 3, 1, 2).float()
    in_stats = None
    if mode == "norm_on_load":
        in_stats = _stats_of(x)
        xin = _inorm_relu(x).float()
    ref = F.conv2d(xin, conv.weight.half().float(), conv.bias.float(), stride=stride, padding=pad)
    res = None
    relu = 0
    if mode == "residual":
        res = torch.randn(B, Ho, Wo, cout, device=dev()).half()
        relu = 1
        ref = F.relu(res.permute(0, 3, 1, 2).float() + F.relu(ref).half().float())
    y = torch.empty(B, Ho, Wo, cout, dtype=torch.float16, device=dev())
    out_stats = torch.zeros(B, cout, 2, device=dev())
    check(lib().vipe_enc_conv(ptr(x), ptr(in_stats) if in_stats is not None else None, ptr(wb[0]), ptr(wb[1]),
                              ptr(res) if res is not None else None, ptr(y), ptr(out_stats), B, H, W, cin, cout, k, stride,
                              relu, 0, -1, stream_ptr(x)), "vipe_enc_conv")
    got = y.permute(0, 3, 1, 2).float()
    scale = float(ref.detach().abs().max())
    assert float((got - ref).abs().max()) < 4e-3 * scale
    if mode != "residual":  # statistics are those of the raw fp16 outputs
        st = _stats_of(y)
        assert torch.allclose(out_stats, st, rtol=2e-4, atol=2e-2)
    # NCHW output + tanh / relu split (the context net's output conv)
    if k == 1 and stride == 1:
        y2 = torch.empty(B, cout, Ho, Wo, dtype=torch.float16, device=dev())
        check(lib().vipe_enc_conv(ptr(x), ptr(in_stats) if in_stats is not None else None, ptr(wb[0]), ptr(wb[1]), None,
                                  ptr(y2), None, B, H, W, cin, cout, k, stride, 0, 1, 128, stream_ptr(x)), "vipe_enc_conv")
        raw = F.conv2d(xin, conv.weight.half().float(), conv.bias.float()).half().float()
        ref2 = torch.cat([raw[:, :128].tanh(), raw[:, 128:].relu()], 1)
        assert float((y2.float() - ref2).abs().max()) < 4e-3 * max(1.0, float(ref2.abs().max()))


def test_encoder_stem_prep_finish_against_torch():
    from vipe_amd.slam.encoders import _pack_stem, normalize_images
    torch.manual_seed(3)
    V, H, W = 2, 50, 90
    img = torch.rand(V, 3, H, W, device=dev())
    x4 = normalize_images(img)
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev()).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=dev()).view(1, 3, 1, 1)
    xn = ((img - mean) / std).half()
    assert torch.equal(x4[..., :3].permute(0, 3, 1, 2), xn) and float(x4[..., 3].abs().max()) == 0
    conv = torch.nn.Conv2d(3, 32, 7, stride=2, padding=3).to(dev())
    wb = _pack_stem(conv)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(V, Ho, Wo, 32, dtype=torch.float16, device=dev())
    stats = torch.zeros(V, 32, 2, device=dev())
    check(lib().vipe_enc_stem(ptr(x4), ptr(wb[0]), ptr(wb[1]), ptr(y), ptr(stats), V, H, W, 0, stream_ptr(x4)), "stem")
    ref = F.conv2d(xn.float(), conv.weight.half().float(), conv.bias.float(), stride=2, padding=3)
    assert tuple(ref.shape[2:]) == (Ho, Wo)
    assert float((y.permute(0, 3, 1, 2).float() - ref).abs().max()) < 4e-3 * float(ref.abs().max())
    assert torch.allclose(stats, _stats_of(y), rtol=2e-4, atol=2e-2)
    # finish: relu(IN(raw)); relu(res + relu(IN(raw))); relu(IN(res) + relu(IN(raw)))
    raw = y
    res = torch.randn_like(y)
    rs = _stats_of(res)
    out = torch.empty_like(y)
    nr = _inorm_relu(raw).permute(0, 2, 3, 1)
    check(lib().vipe_enc_finish(ptr(raw), ptr(stats), None, None, ptr(out), V, Ho * Wo, 32, stream_ptr(raw)), "finish")
    assert float((out.float() - nr.float()).abs().max()) < 2e-3
    check(lib().vipe_enc_finish(ptr(raw), ptr(stats), ptr(res), None, ptr(out), V, Ho * Wo, 32, stream_ptr(raw)), "finish")
    assert float((out.float() - F.relu(res.float() + nr.float())).abs().max()) < 4e-3
    nres = F.instance_norm(res.float().permute(0, 3, 1, 2), eps=1e-5).half().permute(0, 2, 3, 1)
    check(lib().vipe_enc_finish(ptr(raw), ptr(stats), ptr(res), ptr(rs), ptr(out), V, Ho * Wo, 32, stream_ptr(raw)), "finish")
    assert float((out.float() - F.relu(nres.float() + nr.float())).abs().max()) < 4e-3


@pytest.mark.parametrize("tag", ["small", "odd"])
def test_encoders_match_reference_fixture(tag):
    from test_oracle_golden import _encoder_images, _seeded_encoders
    from vipe_amd.slam.encoders import DroidEncoders
    G, fnet, cnet = _seeded_encoders()
    gen = torch.Generator().manual_seed(5)
    for t in ("small", "odd"):  # the fixture's images come from one generator, in this order
        images = _encoder_images(G, t, gen)
        if t == tag:
            break
    enc = DroidEncoders()
    enc.fnet.load_state_dict(fnet.state_dict())
    enc.cnet.load_state_dict(cnet.state_dict())
    enc = enc.to(dev())
    img = images.to(dev()).contiguous()
    fmap = enc.encode_features(img)
    net, inp = enc.encode_context(img)
    for name, got in (("fmap", fmap), ("net", net), ("inp", inp)):
        ref = G[f"{tag}/{name}"]
        assert got.dtype == torch.float16 and tuple(got.shape) == ref.shape
        err = np.abs(got.float().cpu().numpy() - ref).max()
        assert err < 3e-2 * max(1.0, np.abs(ref).max()), (name, err, np.abs(ref).max())


def test_encoder_rejects_cpu_tensors():
    from vipe_amd.slam.encoders import BasicEncoder
    enc = BasicEncoder(128, "instance")
    with pytest.raises(RuntimeError):
        enc.forward_x4(torch.zeros(1, 16, 16, 4, dtype=torch.float16))


def test_motion_filter_score_matches_oracle():
    """MotionFilter.check (motion_filter.py:58-150): first frame -> keyframe; afterwards the mean flow magnitude of one
    update-operator application on the identity grid.  Oracle: fp32 CPU encoders + correlation pyramid / lookup +
    update operator restatements.  fp16 device path vs fp32 oracle: 3e-2 relative on the score."""
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
    V, H, W = 1, 96, 128
    img0 = torch.rand(V, 3, H, W, generator=gen)
    img1 = (img0 + 0.1 * torch.rand(V, 3, H, W, generator=gen)).clamp(0, 1)
    mf = MotionFilter(dn, thresh=1e9, device=dev())
    assert mf.check(img0.to(dev()).contiguous(), None) is True
    assert mf.check(img1.to(dev()).contiguous(), None) is False
    # masked variant: the left half of the keyframe is invalid
    mask = torch.zeros(V, H // 8, W // 8, dtype=torch.bool)
    mask[:, :, : W // 16] = True
    mf_m = MotionFilter(dn, thresh=1e9, device=dev())
    mf_m.check(img0.to(dev()).contiguous(), mask.to(dev()))
    mf_m.check(img1.to(dev()).contiguous(), None)
    with torch.no_grad():
        f0, f1 = oenc.encode_features(sd_f, img0), oenc.encode_features(sd_f, img1)
        net, inp = oenc.encode_context(sd_c, img0)
        ht, wd = H // 8, W // 8
        pyr = [l.numpy() for l in ocorr.corr_pyramid(f0[None], f1[None])]
        coords0 = MotionFilter.coords_grid(ht, wd)[None, None].repeat(1, V, 1, 1, 1).numpy()
        corr = torch.from_numpy(ocorr.corr_lookup(pyr, coords0, 3))
        _, delta, _ = oum.update_forward(sd_u, net[None], inp[None], corr, torch.zeros(1, V, 4, ht, wd))
        flow = delta.norm(dim=-1)[0]
        ref = float(flow.mean([1, 2]).min())
        fw = (~mask).float()
        ref_m = float(((flow * fw).mean([1, 2]) / (fw.mean([1, 2]) + 1e-6)).min())
    assert abs(mf.last_score - ref) < 3e-2 * max(ref, 1e-3), (mf.last_score, ref)
    assert abs(mf_m.last_score - ref_m) < 3e-2 * max(ref_m, 1e-3), (mf_m.last_score, ref_m)
    # a zero threshold keeps the frame and refreshes the keyframe features
    mf0 = MotionFilter(dn, thresh=0.0, device=dev())
    mf0.check(img0.to(dev()).contiguous(), None)
    old = mf0.f_fmap.clone()
    assert mf0.check(img1.to(dev()).contiguous(), None) is True and mf0.last_kf_frame_idx == 1
    assert not torch.equal(old, mf0.f_fmap)
