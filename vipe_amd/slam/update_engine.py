"""Execution engine of the flow-update operator (UpdateModule.forward, droid_net.py:467-499) on MI355X.

Data layout: every activation is NHWC fp16 ([E, h, w, C], channels innermost) so that a 3x3 tap reads
C contiguous halves (256 B at C=128) and the implicit-GEMM K dimension is contiguous for MFMA fragments.
Concatenations of the reference (`cat[net, inp, corr_feat, flow_feat]`, droid_net.py:388-389) are never
materialised separately: producers write straight into channel slices of one [E,h,w,448] buffer.

Backends
  * "hip"   : hand-written MFMA implicit-GEMM convolutions of libvipe_amd.so (vipe_conv2d_nhwc_f16) with
              fused bias / activation / per-image additive term.
  * "miopen": the same dataflow with torch.nn.functional.conv2d in fp16 channels_last (MIOpen / hipBLASLt).
              Kept as the A/B baseline for the hand-written kernels; selected only explicitly or through
              VIPE_AMD_CONV=miopen.
"""

import os

import torch
import torch.nn.functional as F

from .._lib import check, lib, ptr, stream_ptr

ACT = {"none": 0, "relu": 1, "sigmoid": 2, "tanh": 3}


class _ConvW:
    """One convolution's parameters in both layouts."""

    def __init__(self, conv, device, cout_slice=None):
        w = conv.weight.detach()
        b = conv.bias.detach()
        if cout_slice is not None:
            w, b = w[cout_slice], b[cout_slice]
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.w_oihw = w.to(device=device, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        self.bias16 = b.to(device=device, dtype=torch.float16)
        self.bias32 = b.to(device=device, dtype=torch.float32).contiguous()
        self.w_packed = None  # [KH*KW, Cin, Cout] fp16, built lazily for the hip backend


class UpdateEngine:
    def __init__(self, module, device, backend=None):
        self.device = device
        self.backend = backend or os.environ.get("VIPE_AMD_CONV", "hip")
        m = module
        d = device
        self.w = {
            "corr0": _ConvW(m.corr_encoder[0], d), "corr2": _ConvW(m.corr_encoder[2], d),
            "flow0": _ConvW(m.flow_encoder[0], d), "flow2": _ConvW(m.flow_encoder[2], d),
            "weight0": _ConvW(m.weight[0], d), "weight2": _ConvW(m.weight[2], d),
            "delta0": _ConvW(m.delta[0], d), "delta2": _ConvW(m.delta[2], d),
            "convz": _ConvW(m.gru.convz, d), "convr": _ConvW(m.gru.convr, d), "convq": _ConvW(m.gru.convq, d),
            "w": _ConvW(m.gru.w, d), "convz_glo": _ConvW(m.gru.convz_glo, d), "convr_glo": _ConvW(m.gru.convr_glo, d),
            "convq_glo": _ConvW(m.gru.convq_glo, d),
            "agg1": _ConvW(m.agg.conv1, d), "agg2": _ConvW(m.agg.conv2, d), "eta": _ConvW(m.agg.eta[0], d),
            "upmask": _ConvW(m.agg.upmask[0], d),
        }
        if self.backend == "hip":
            for cw in self.w.values():
                self._pack(cw)

    # ------------------------------------------------------------------ convolution primitive
    def _pack(self, cw):
        packed = torch.empty((cw.kh * cw.kw, cw.cin, cw.cout), dtype=torch.float16, device=self.device)
        src = cw.w_oihw.contiguous()  # plain OIHW for the packer
        check(lib().vipe_conv_pack_weights(ptr(src), ptr(packed), cw.cout, cw.cin, cw.kh, cw.kw, 0, stream_ptr(src)),
              "conv_pack_weights")
        cw.w_packed = packed

    def conv(self, x, name, act="none", out=None, cout_off=0, cin_off=0, extra=None):
        """x [B,H,W,Ctot] fp16 NHWC (reads channels cin_off:cin_off+Cin) -> out[..., cout_off:cout_off+Cout]."""
        cw = self.w[name]
        B, H, W, ctot = x.shape
        if out is None:
            out = torch.empty((B, H, W, cw.cout), dtype=torch.float16, device=x.device)
        if self.backend == "hip":
            check(lib().vipe_conv2d_nhwc_f16(ptr(x), ptr(cw.w_packed), ptr(cw.bias32), ptr(extra), ptr(out), B, H, W,
                                             cw.cin, ctot, cin_off, cw.cout, out.shape[-1], cout_off, cw.kh, cw.kw,
                                             ACT[act], stream_ptr(x)), "conv2d_nhwc_f16")
            return out
        xin = x[..., cin_off:cin_off + cw.cin].permute(0, 3, 1, 2)  # NCHW view of NHWC memory (channels_last)
        y = F.conv2d(xin, cw.w_oihw, cw.bias16, padding=cw.kh // 2)
        if extra is not None:
            y = y + extra.to(torch.float16)[:, :, None, None]
        y = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](y)
        out[..., cout_off:cout_off + cw.cout] = y.permute(0, 2, 3, 1)
        return out

    # ------------------------------------------------------------------ the operator
    @torch.no_grad()
    def forward(self, net, inp, corr, flow=None, ix=None, skip_upmask=False, n_src=None):
        """Reference signature and return structure (droid_net.py:467-499); tensors arrive NCHW like the reference
        ([1,E,C,h,w]); NHWC buffers are used internally."""
        batch, num, ch, ht, wd = net.shape
        E = batch * num
        dev = net.device
        f16 = torch.float16

        def nhwc(t, c):
            return t.reshape(E, c, ht, wd).to(f16).permute(0, 2, 3, 1).contiguous()

        # hx = [net(128) | inp(128) | corr_feat(128) | flow_feat(64)]  (droid_net.py:388-389)
        hx = torch.empty((E, ht, wd, 448), dtype=f16, device=dev)
        hx[..., 0:128] = net.reshape(E, 128, ht, wd).permute(0, 2, 3, 1)
        hx[..., 128:256] = inp.reshape(E, 128, ht, wd).permute(0, 2, 3, 1)
        corr_n = nhwc(corr, 196)
        if flow is None:
            flow_n = torch.zeros((E, ht, wd, 4), dtype=f16, device=dev)
        else:
            flow_n = nhwc(flow, 4)

        c1 = self.conv(corr_n, "corr0", "relu")
        self.conv(c1, "corr2", "relu", out=hx, cout_off=256)
        f1 = self.conv(flow_n, "flow0", "relu")
        self.conv(f1, "flow2", "relu", out=hx, cout_off=384)

        # global context (droid_net.py:392-393): glo = mean_hw(sigmoid(w(net)) * net)
        g = self.conv(hx, "w", "sigmoid")  # reads channels 0:128
        glo = (g.float() * hx[..., 0:128].float()).mean(dim=(1, 2))  # [E,128] fp32
        glo16 = glo.to(f16)

        def glo_term(name):
            cw = self.w[name]
            return (glo16 @ cw.w_oihw.reshape(cw.cout, cw.cin).t() + cw.bias16).float().contiguous()

        z = self.conv(hx, "convz", "sigmoid", extra=glo_term("convz_glo"))
        r = self.conv(hx, "convr", "sigmoid", extra=glo_term("convr_glo"))
        rhx = hx.clone()
        rhx[..., 0:128] = r * hx[..., 0:128]
        q = self.conv(rhx, "convq", "tanh", extra=glo_term("convq_glo"))
        net_n = (1 - z) * hx[..., 0:128] + z * q  # [E,h,w,128] fp16

        d1 = self.conv(net_n, "delta0", "relu")
        delta = self.conv(d1, "delta2", "none")
        w1 = self.conv(net_n, "weight0", "relu")
        weight = self.conv(w1, "weight2", "sigmoid")
        delta = delta.view(batch, num, ht, wd, -1)[..., :2].contiguous()
        weight = weight.view(batch, num, ht, wd, -1)[..., :2].contiguous()
        net_out = net_n.permute(0, 3, 1, 2).reshape(batch, num, 128, ht, wd)
        if ix is None:
            return net_out, delta, weight

        # GraphAgg (droid_net.py:414-429)
        a = self.conv(net_n, "agg1", "relu")  # [E,h,w,128]
        if n_src is None:  # the reference syncs here too (scatter.py:40: int(index.max()) + 1)
            n_src = int(ix.max().item()) + 1 if ix.numel() else 0
        ixd = ix.to(dev)
        acc = torch.zeros((n_src, ht, wd, 128), dtype=torch.float32, device=dev).index_add_(0, ixd, a.float())
        cnt = torch.zeros(n_src, dtype=torch.float32, device=dev).index_add_(0, ixd, torch.ones(E, device=dev))
        a = (acc / cnt.clamp(min=1).view(-1, 1, 1, 1)).to(f16)
        a = self.conv(a, "agg2", "relu")
        eta = F.softplus(self.conv(a, "eta", "none").float()).view(batch, n_src, ht, wd)
        upmask = None
        if not skip_upmask:
            upmask = self.conv(a, "upmask", "none").permute(0, 3, 1, 2).reshape(batch, n_src, 8 * 8 * 9, ht, wd)
        return net_out, delta, weight, 0.01 * eta, upmask
