"""Frame encoders and the motion filter - host-side mirror of `BasicEncoder` (vipe/slam/networks/droid_net.py:290-370),
`DroidNet.encode_features / encode_context` (:510-527) and `MotionFilter.check`
(vipe/slam/components/motion_filter.py:58-150).  SURVEY 8(f) row 2.

`BasicEncoder` keeps the reference's constructor arguments, state-dict keys and default initialisation (so `fnet.*` /
`cnet.*` checkpoints load with `load_state_dict`); its forward runs the HIP kernels of `csrc/encoder.hip` through the
C ABI (`vipe_enc_*`): NHWC fp16 activations, instance-norm statistics produced by the convolution that writes a
tensor and applied by the one that reads it.  No torch fallback: CPU tensors raise."""
import torch
import torch.nn as nn

from .._lib import check, lib, ptr, require, stream_ptr

DIM = 32


class _Block(nn.Module):
    """parameter holder with ResidualBlock's layout (droid_net.py:179-219)"""

    def __init__(self, in_planes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.stride = stride
        self.downsample = None if stride == 1 else nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride))


def _pack_conv(conv):
    """[Cout,Cin,k,k] -> [k*k][Cin/32][Cout][32] fp16"""
    w = conv.weight.detach()
    co, ci, kh, kw = w.shape
    w = w.permute(2, 3, 1, 0).reshape(kh * kw, ci // 32, 32, co).permute(0, 1, 3, 2)
    return w.contiguous().half(), conv.bias.detach().float().contiguous()


def _pack_stem(conv):
    """[32,3,7,7] -> [7][32][32] fp16 with k = tap*4 + c"""
    w = conv.weight.detach()
    co = w.shape[0]
    k = torch.zeros(56, 4, co, dtype=w.dtype, device=w.device)
    k[:49, :3] = w.permute(2, 3, 1, 0).reshape(49, 3, co)
    return k.reshape(7, 32, co).permute(0, 2, 1).contiguous().half(), conv.bias.detach().float().contiguous()


class BasicEncoder(nn.Module):
    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0, multidim=False):
        super().__init__()
        require(norm_fn in ("instance", "none"), "the SLAM encoders use norm_fn 'instance' (fnet) or 'none' (cnet)")
        require(not multidim and dropout == 0.0, "multidim / dropout variants are not used by DroidNet")
        self.norm_fn, self.output_dim = norm_fn, output_dim
        self.conv1 = nn.Conv2d(3, DIM, kernel_size=7, stride=2, padding=3)
        self.layer1 = nn.Sequential(_Block(DIM, DIM, 1), _Block(DIM, DIM, 1))
        self.layer2 = nn.Sequential(_Block(DIM, 2 * DIM, 2), _Block(2 * DIM, 2 * DIM, 1))
        self.layer3 = nn.Sequential(_Block(2 * DIM, 4 * DIM, 2), _Block(4 * DIM, 4 * DIM, 1))
        self.conv2 = nn.Conv2d(4 * DIM, output_dim, kernel_size=1)
        for m in self.modules():  # droid_net.py:340-346
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._packed = None

    # ---- packed weights (rebuilt after load_state_dict / .to())
    def _pack(self, device):
        if self._packed is not None and self._packed["dev"] == device:
            return self._packed
        P = {"dev": device, "stem": _pack_stem(self.conv1.to(device)), "out": _pack_conv(self.conv2.to(device))}
        for li in (1, 2, 3):
            for bi, blk in enumerate(getattr(self, f"layer{li}")):
                blk = blk.to(device)
                P[(li, bi, "conv1")] = _pack_conv(blk.conv1)
                P[(li, bi, "conv2")] = _pack_conv(blk.conv2)
                if blk.downsample is not None:
                    P[(li, bi, "ds")] = _pack_conv(blk.downsample[0])
        self._packed = P
        return P

    def load_state_dict(self, *a, **k):
        self._packed = None
        return super().load_state_dict(*a, **k)

    @staticmethod
    def _conv(x, wb, cout, k, stride, st, in_stats=None, res=None, out_stats=None, relu=0, nchw=0, tanh_split=-1):
        B, H, W, cin = x.shape
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = torch.empty((B, cout, Ho, Wo) if nchw else (B, Ho, Wo, cout), dtype=torch.float16, device=x.device)
        check(lib().vipe_enc_conv(ptr(x), ptr(in_stats) if in_stats is not None else None, ptr(wb[0]), ptr(wb[1]),
                                  ptr(res) if res is not None else None, ptr(y),
                                  ptr(out_stats) if out_stats is not None else None, B, H, W, cin, cout, k, stride, relu,
                                  nchw, tanh_split, st), "vipe_enc_conv")
        return y

    def forward_x4(self, x4, tanh_split=-1):
        """x4 [n,H,W,4] fp16 normalised image (vipe_enc_prep) -> [n,output_dim,H/8,W/8] fp16 (NCHW)."""
        require(x4.is_cuda and x4.dtype == torch.float16 and x4.is_contiguous() and x4.shape[-1] == 4,
                "BasicEncoder runs on the HIP device over a normalised NHWC4 fp16 image")
        if x4.device.index is not None and x4.device.index != torch.cuda.current_device():
            with torch.cuda.device(x4.device):  # the launches below share one stream handle: make its device current
                return self.forward_x4(x4, tanh_split)
        P = self._pack(x4.device)
        st = stream_ptr(x4)
        n, H, W, _ = x4.shape
        inorm = self.norm_fn == "instance"
        # one zeroed arena for every statistics buffer of this pass: 14 tensors of at most 128 channels
        arena = torch.zeros((16, n, 128, 2), dtype=torch.float32, device=x4.device) if inorm else None
        slot = [0]

        def stats(c):
            if not inorm:
                return None
            s = arena[slot[0]].view(-1)[: n * c * 2]
            slot[0] += 1
            return s

        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((n, Ho, Wo, DIM), dtype=torch.float16, device=x4.device)
        s0 = stats(DIM)
        check(lib().vipe_enc_stem(ptr(x4), ptr(P["stem"][0]), ptr(P["stem"][1]), ptr(y), ptr(s0) if inorm else None, n, H,
                                  W, 0 if inorm else 1, st), "vipe_enc_stem")
        if inorm:
            x = torch.empty_like(y)
            check(lib().vipe_enc_finish(ptr(y), ptr(s0), None, None, ptr(x), n, Ho * Wo, DIM, st), "vipe_enc_finish")
        else:
            x = y
        for li, planes in ((1, DIM), (2, 2 * DIM), (3, 4 * DIM)):
            for bi, blk in enumerate(getattr(self, f"layer{li}")):
                stride = blk.stride
                if inorm:
                    s1, s2 = stats(planes), stats(planes)
                    r1 = self._conv(x, P[(li, bi, "conv1")], planes, 3, stride, st, out_stats=s1)
                    r2 = self._conv(r1, P[(li, bi, "conv2")], planes, 3, 1, st, in_stats=s1, out_stats=s2)
                    res, rs = x, None
                    if stride != 1:
                        rs = stats(planes)
                        res = self._conv(x, P[(li, bi, "ds")], planes, 1, stride, st, out_stats=rs)
                    out = torch.empty_like(r2)
                    check(lib().vipe_enc_finish(ptr(r2), ptr(s2), ptr(res), ptr(rs) if rs is not None else None, ptr(out),
                                                n, r2.shape[1] * r2.shape[2], planes, st), "vipe_enc_finish")
                    x = out
                else:
                    r1 = self._conv(x, P[(li, bi, "conv1")], planes, 3, stride, st, relu=1)
                    res = x if stride == 1 else self._conv(x, P[(li, bi, "ds")], planes, 1, stride, st)
                    x = self._conv(r1, P[(li, bi, "conv2")], planes, 3, 1, st, res=res, relu=1)
        return self._conv(x, P["out"], self.output_dim, 1, 1, st, nchw=1, tanh_split=tanh_split)

    def forward(self, x):
        """x [b,n,3,H,W] normalised fp32/fp16 image -> [b,n,output_dim,H/8,W/8] (droid_net.py:352-370)"""
        b, n, c, H, W = x.shape
        x4 = torch.zeros((b * n, H, W, 4), dtype=torch.float16, device=x.device)
        x4[..., :3] = x.reshape(b * n, c, H, W).permute(0, 2, 3, 1)
        out = self.forward_x4(x4)
        return out.view(b, n, *out.shape[1:])


def normalize_images(images):
    """[V,3,H,W] fp32 RGB in [0,1] on the device -> [V,H,W,4] fp16 (droid_net.py:512-516)"""
    require(images.is_cuda and images.dtype == torch.float32 and images.is_contiguous() and images.shape[1] == 3,
            "images must be a contiguous fp32 [V,3,H,W] device tensor")
    V, _, H, W = images.shape
    x4 = torch.empty((V, H, W, 4), dtype=torch.float16, device=images.device)
    check(lib().vipe_enc_prep(ptr(images), ptr(x4), V, H, W, stream_ptr(images)), "vipe_enc_prep")
    return x4


class DroidEncoders(nn.Module):
    """fnet + cnet with DroidNet's two entry points (droid_net.py:503-527); `update` is attached by the caller."""

    def __init__(self):
        super().__init__()
        self.fnet = BasicEncoder(output_dim=128, norm_fn="instance")
        self.cnet = BasicEncoder(output_dim=256, norm_fn="none")

    @torch.no_grad()
    def encode_features(self, images, x4=None):
        return self.fnet.forward_x4(normalize_images(images) if x4 is None else x4)

    @torch.no_grad()
    def encode_context(self, images, x4=None):
        out = self.cnet.forward_x4(normalize_images(images) if x4 is None else x4, tanh_split=128)
        return out[:, :128], out[:, 128:]
