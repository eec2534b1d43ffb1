"""Execution engine of the flow-update operator (UpdateModule.forward, droid_net.py:467-499) on MI355X.

Data layout: every activation is NHWC fp16 ([E, h, w, C], channels innermost) so that a 3x3 tap reads
C contiguous halves and the implicit-GEMM K dimension is contiguous for MFMA fragments.  The concatenations
of the reference (`cat[net, inp, corr_feat, flow_feat]`, droid_net.py:388-389) are never materialised: the
GRU convolutions read channels [0,128) from the hidden state (or r*net) and [128,448) from one
[E,h,w,320] buffer that the encoders write into.

13 launches per update, all hand-written MFMA implicit-GEMM convolutions (csrc/conv_mfma.hip) with fused
epilogues:
    corr0 (1x1 200->128, relu) -> corr2 (3x3, relu)            -> xbuf[128:256]
    flow0 (7x7 4->128, relu)   -> flow2 (3x3 128->64, relu)    -> xbuf[256:320]
    w     (1x1, sigmoid * net, summed over pixels)             -> glo_sum        [GLO epilogue]
    convz|convr (3x3 448->256): z, r*net                                          [ZR epilogue]
    convq (3x3 448->128): net' = (1-z) net + z tanh(.)                            [Q epilogue]
    delta0|weight0|agg1 (3x3 128->384, relu) -> delta2|weight2 (3x3 256->4)       [HEADS epilogue]
    agg2 (3x3, relu) -> eta (3x3 128->1, 0.01 softplus)                           [ETA epilogue]

There is no other backend: the torch / MIOpen formulation used as the A/B baseline of these kernels lives in
scratch/miopen_ab.py (diagnostic, not importable from the package).
"""

import ctypes
import os

import torch

from .._lib import check, lib, parse_struct, ptr, stream_ptr

UpdateWeights = parse_struct("vipe_update_weights")   # include/vipe_amd.h: field order / types come from the header
UpdateBuffers = parse_struct("vipe_update_buffers")
GateStateJob = parse_struct("vipe_gate_state_job")

ACT = {"none": 0, "relu": 1, "sigmoid": 2, "tanh": 3}
EPI = {"plain": 0, "glo": 1, "zr": 2, "q": 3, "heads": 4, "eta": 5, "partial": 6}
ACCINIT_F32 = 0x100  # VIPE_CONV_ACCINIT_F32
CORR_CH = 200  # 196 correlation channels padded to a multiple of 8 (zeros)


class _Packed:
    """Weights of one (possibly fused) convolution packed for conv_mfma: [K_pad/64][Cout_pad][64] fp16."""

    def __init__(self, w_oihw, bias, device):
        w = w_oihw.detach().to(device=device, dtype=torch.float16).contiguous()
        self.cout, self.cin, self.kh, self.kw = w.shape
        cp, cinp, kp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().vipe_conv_packed_dims(self.cout, self.cin, self.kh, self.kw, ctypes.addressof(cp),
                                          ctypes.addressof(cinp), ctypes.addressof(kp)), "conv_packed_dims")
        self.packed = torch.empty(kp.value * cp.value, dtype=torch.float16, device=device)
        check(lib().vipe_conv_pack_weights(ptr(w), ptr(self.packed), self.cout, self.cin, self.kh, self.kw, 0,
                                           stream_ptr(w)), "conv_pack_weights")
        self.bias = bias.detach().to(device=device, dtype=torch.float32).contiguous()
        self.flops_per_pixel = 2.0 * self.cout * self.cin * self.kh * self.kw


def segment_csr(ix, n_src):
    """(order, rowptr) int32 device tensors grouping the edges by source slot (stable)."""
    order = torch.argsort(ix, stable=True).to(torch.int32).contiguous()
    counts = torch.bincount(ix, minlength=n_src)
    rowptr = torch.zeros(n_src + 1, dtype=torch.int32, device=ix.device)
    rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return order, rowptr


class UpdateEngine:
    def __init__(self, module, device):
        self.device = device
        self.m = module
        self._bufs = {}
        # second stream of the natively sequenced operator (vipe_update_buffers.side_stream): worth its four event
        # operations only when the kernels are long enough, i.e. for large edge sets
        self._op_side = None
        self.op_side_min_edges = int(os.environ.get("VIPE_AMD_OP_SIDE_MIN_EDGES", "64"))
        # A staged gate state (hidden_gate_state / gate_state_job) lives in THIS engine's scratch buffers (pzr, extra, glo),
        # and one engine serves every FactorGraph of an UpdateModule (frontend, backend, infill): any later call that
        # writes those buffers bumps the generation, and a state of an older generation no longer matches
        self._gate_gen = 0
        self._pack_all()

    # ------------------------------------------------------------------ weights
    def _pack_all(self):
        m, d = self.m, self.device
        f16 = torch.float16

        def wb(conv):
            return conv.weight.detach().to(d, f16), conv.bias.detach().to(d, torch.float32)

        w, b = wb(m.corr_encoder[0])
        w = torch.cat([w, torch.zeros(128, CORR_CH - 196, 1, 1, device=d, dtype=f16)], 1)
        self.corr0 = _Packed(w, b, d)
        self.corr2 = _Packed(*wb(m.corr_encoder[2]), d)
        self.flow0 = _Packed(*wb(m.flow_encoder[0]), d)
        self.flow2 = _Packed(*wb(m.flow_encoder[2]), d)
        self.gw = _Packed(*wb(m.gru.w), d)
        wz, bz = wb(m.gru.convz)
        wr, br = wb(m.gru.convr)
        self.zr = _Packed(torch.cat([wz, wr], 0), torch.cat([bz, br], 0), d)
        wq, bq = wb(m.gru.convq)
        self.q = _Packed(wq, bq, d)
        # the same gates split by input channel group ([net | inp | corr | flow], droid_net.py:395-399): everything but
        # `inp` per iteration, the `inp` part once per edge (gate_context)
        rest = list(range(0, 128)) + list(range(256, 448))
        self.zr_s = _Packed(torch.cat([wz, wr], 0)[:, rest].contiguous(), torch.cat([bz, br], 0), d)
        self.q_s = _Packed(wq[:, rest].contiguous(), bq, d)
        self.gates_inp = _Packed(torch.cat([wz, wr, wq], 0)[:, 128:256].contiguous(), torch.zeros(384, device=d), d)
        # ... and z|r once more split into its hidden-state part and its (corr | flow) part (hidden_gate_state)
        self.zr_n = _Packed(torch.cat([wz, wr], 0)[:, 0:128].contiguous(), torch.cat([bz, br], 0), d)
        self.zr_x = _Packed(torch.cat([wz, wr], 0)[:, 256:448].contiguous(), torch.cat([bz, br], 0), d)
        wd0, bd0 = wb(m.delta[0])
        ww0, bw0 = wb(m.weight[0])
        wa1, ba1 = wb(m.agg.conv1)
        self.heads0 = _Packed(torch.cat([wd0, ww0, wa1], 0), torch.cat([bd0, bw0, ba1], 0), d)
        wd2, bd2 = wb(m.delta[2])
        ww2, bw2 = wb(m.weight[2])
        h2 = torch.zeros(4, 256, 3, 3, device=d, dtype=f16)
        h2[0:2, 0:128] = wd2[:2]
        h2[2:4, 128:256] = ww2[:2]
        self.heads2 = _Packed(h2, torch.cat([bd2[:2], bw2[:2]], 0), d)
        self.agg2 = _Packed(*wb(m.agg.conv2), d)
        self.eta = _Packed(*wb(m.agg.eta[0]), d)
        self.upmask = _Packed(*wb(m.agg.upmask[0]), d)
        # global-context 1x1s applied to the pooled vector: one [128, 384] matrix (z | r | q)
        g = m.gru
        self.glo_w = torch.cat([g.convz_glo.weight, g.convr_glo.weight, g.convq_glo.weight], 0).detach().to(d).float() \
            .reshape(384, 128).t().contiguous()
        self.glo_b = torch.cat([g.convz_glo.bias, g.convr_glo.bias, g.convq_glo.bias], 0).detach().to(d).float()
        # descriptor of all of the above for the natively sequenced operator (vipe_update_operator)
        wd_ = UpdateWeights()
        for name, pk in dict(corr0=self.corr0, corr2=self.corr2, flow0=self.flow0, flow2=self.flow2, gw=self.gw, zr=self.zr,
                             q=self.q, heads0=self.heads0, heads2=self.heads2, agg2=self.agg2, eta=self.eta).items():
            setattr(wd_, name + "_w", pk.packed.data_ptr())
            setattr(wd_, name + "_b", pk.bias.data_ptr())
        wd_.zr_s_w, wd_.q_s_w = self.zr_s.packed.data_ptr(), self.q_s.packed.data_ptr()
        wd_.zr_n_w, wd_.zr_x_w = self.zr_n.packed.data_ptr(), self.zr_x.packed.data_ptr()
        wd_.glo_wT, wd_.glo_b = self.glo_w.data_ptr(), self.glo_b.data_ptr()
        self._wdesc = wd_
        self._bdesc = {}

    # ------------------------------------------------------------------ launches
    def _conv(self, pk, x0, x0_coff, B, H, W, y=None, y_coff=0, act="none", mode="plain", x1=None, x1_coff=0,
              split=None, extra=None, extra_off=0, y2=None, y2_coff=0, net=None, net_coff=0, z=None, fout=None,
              cin=None, accinit=None, ai_coff=0):
        cin = pk.cin if cin is None else cin
        split = cin if split is None else split
        check(lib().vipe_conv2d_fused(
            ptr(x0), x0.shape[-1], x0_coff, ptr(x1), x1.shape[-1] if x1 is not None else 0, x1_coff, split,
            ptr(pk.packed), ptr(pk.bias), ptr(extra), extra.shape[-1] if extra is not None else 0, extra_off,
            ptr(y), (y if y is not None else fout).shape[-1] if (y is not None or mode == "partial") else 0, y_coff,
            ptr(y2), y2.shape[-1] if y2 is not None else 0, y2_coff,
            ptr(net), net.shape[-1] if net is not None else 0, net_coff, ptr(z), ptr(fout), ptr(accinit),
            accinit.shape[-1] if accinit is not None else 0, ai_coff, B, H, W, cin, pk.cout,
            pk.kh, pk.kw, ACT[act],
            EPI[mode] | (ACCINIT_F32 if accinit is not None and accinit.dtype == torch.float32 else 0),
            stream_ptr(x0)), "conv2d_fused")

    @staticmethod
    def supports_gate_split(ht, wd):
        """Initial accumulators are implemented by the tile convolutions (csrc/conv_mfma.hip): the 4 x 64 tiling for
        grids made of such tiles, the flat tiling for every other grid up to 126 columns - i.e. every grid the
        reference's resize to 384 x 512 pixels of area produces (vipe/slam/system.py:46-59: 41 x 73 for 16:9 video)."""
        return (wd % 64 == 0 and ht % 4 == 0) or wd <= 126

    def gate_context(self, xbuf, out=None):
        """The part of the three GRU gate convolutions that only depends on the context features `inp`
        (xbuf[..., 0:128]; constant for the lifetime of an edge): [E,h,w,384] fp16 = conv3x3(inp; W_{z|r|q}[:, 128:256]).
        Handed back to `forward_nhwc(pgate=...)`, where it is the initial value of the gate accumulators, so that the
        per-iteration gate convolutions run over 320 instead of 448 input channels (19 % of the operator's FLOPs).
        `out`: a contiguous [E,h,w,384] fp16 destination (the tail of a store) instead of a fresh tensor."""
        E, H, W, _ = xbuf.shape
        pg = out if out is not None else torch.empty((E, H, W, 384), dtype=torch.float16, device=xbuf.device)
        self._conv(self.gates_inp, xbuf, 0, E, H, W, y=pg, act="none", cin=128)
        return pg

    @torch.no_grad()
    def hidden_gate_state(self, net, pgate, native=True, parts=3, n_staged=None):
        """Everything of the GRU gates that depends on the hidden state `net` [E,h,w,128] alone, on the CURRENT stream:
        the three global-context terms (`extra` [E,384]) and pzr [E,h,w,256] f32 = pgate[..., :256] +
        conv3x3(net; W_{z|r}[:, 0:128]).  `forward_nhwc(net, ..., gate_state=<the returned dict>)` then skips the
        global-context stage and runs the z|r convolution over the (corr | flow) channels only - same result up to
        fp32 summation order.  `FactorGraph.update` issues this for the NEW hidden state on a second stream right after
        the operator, so that it runs while the dense BA (one busy workgroup for most of its time) has the chip."""
        E, H, W, _ = net.shape
        Es = E if n_staged is None else max(1, min(E, int(n_staged)))  # the z|r part is staged for the first Es edges
        self._gate_gen += 1
        glo = self._buf("glo", (E, 128), torch.float32)
        extra = self._buf("extra", (E, 384), torch.float32)
        pzr = self._buf("pzr", (Es, H, W, 256), torch.float32)
        if native:
            b = UpdateBuffers()
            b.H, b.W = H, W
            b.pgate, b.pzr, b.glo, b.extra = pgate.data_ptr(), pzr.data_ptr(), glo.data_ptr(), extra.data_ptr()
            for part, n in ((1, E), (2, Es)):
                if parts & part:
                    b.E = n
                    check(lib().vipe_update_gate_state(ctypes.addressof(self._wdesc), ctypes.addressof(b), ptr(net), part,
                                                       stream_ptr(net)), "update_gate_state")
        else:
            if parts & 1:
                glo.zero_()
                self._conv(self.gw, net, 0, E, H, W, mode="glo", net=net, fout=glo)
                check(lib().vipe_glo_context(ptr(glo), ptr(self.glo_w), ptr(self.glo_b), ptr(extra), E, H * W,
                                             stream_ptr(glo)), "glo_context")
            if parts & 2:
                self._conv(self.zr_n, net, 0, Es, H, W, mode="partial", fout=pzr, accinit=pgate, ai_coff=0)
        return dict(net_ptr=net.data_ptr(), pgate_ptr=pgate.data_ptr(), shape=tuple(net.shape), pzr=pzr, extra=extra,
                    n_staged=Es, gen=self._gate_gen)

    def gate_state_job(self, net, pgate, fractions=None, n_staged=None):
        """The z|r part of `hidden_gate_state(net, pgate)` (parts = 2), not launched: (gate_state dict, overlap triple) for
        `slam_ext.dense_ba(overlap=...)` / `GraphBuffer.bundle_adjustment(overlap=...)` - the library enqueues it in
        pieces (`vipe_update_gate_state_piece`) on the given stream while the BA's solve kernels run; `fractions`:
        share of the edges per piece (default: even).  The dict is valid once that BA call has returned, the stream has
        been joined AND `hidden_gate_state(net, pgate, parts=1)` has run; it keeps the descriptors alive."""
        E_all, H, W, _ = net.shape
        E = E_all if n_staged is None else max(1, min(E_all, int(n_staged)))  # the job covers the first E edges
        self._gate_gen += 1
        glo = self._buf("glo", (E_all, 128), torch.float32)
        extra = self._buf("extra", (E_all, 384), torch.float32)
        pzr = self._buf("pzr", (E, H, W, 256), torch.float32)
        b = UpdateBuffers()
        b.E, b.H, b.W = E, H, W
        b.pgate, b.pzr, b.glo, b.extra = pgate.data_ptr(), pzr.data_ptr(), glo.data_ptr(), extra.data_ptr()
        job = GateStateJob()
        job.weights, job.buffers, job.net = ctypes.addressof(self._wdesc), ctypes.addressof(b), net.data_ptr()
        bounds = None
        if fractions:
            cum, acc = [0], 0.0
            for f in fractions[:-1]:
                acc += f
                cum.append(min(E, int(round(E * acc / sum(fractions)))))
            cum.append(E)
            bounds = (ctypes.c_int * len(cum))(*cum)
            job.bounds, job.n_bounds = ctypes.addressof(bounds), len(cum)
        fn = ctypes.cast(lib().vipe_update_gate_state_piece, ctypes.c_void_p).value
        gs = dict(net_ptr=net.data_ptr(), pgate_ptr=pgate.data_ptr(), shape=tuple(net.shape), pzr=pzr, extra=extra,
                  n_staged=E, keep=(b, job, bounds), gen=self._gate_gen)
        return gs, (lambda stream: (stream.cuda_stream, fn, ctypes.addressof(job)))

    def gate_state_matches(self, gs, net, pgate):
        return (gs is not None and pgate is not None and gs["net_ptr"] == net.data_ptr() and
                gs["pgate_ptr"] == pgate.data_ptr() and gs["shape"] == tuple(net.shape) and gs.get("gen") == self._gate_gen)

    def _buf(self, name, shape, dtype=torch.float16):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t

    @torch.no_grad()
    def forward_nhwc(self, net, xbuf, corr, motn, ix=None, n_src=None, net_out=None, want_upmask=False, csr=None,
                     pgate=None, native=True, gate_state=None):
        """The operator on channels-last state.

        net  [E,h,w,128] f16 hidden state;  xbuf [E,h,w,320] f16 with the context features `inp` in channels
        [0,128) (channels [128,320) are scratch, overwritten);  corr [E,h,w,200] f16 (196 + zero pad);
        motn [E,h,w,4] f16;  ix [E] int64 -> source slot.  Returns (net' [E,h,w,128] f16, dw [E,h,w,4] f32 =
        (delta_x, delta_y, weight_x, weight_y), eta [n_src,h,w] f32 or None, upmask or None).

        `native` (default): ONE library call sequences the whole operator (`vipe_update_operator`); native=False issues
        the same kernels one by one from here - the instrumentable form (bench.py brackets single launches with events),
        checked equal to the native one by tests/test_gpu_parity.py.  `gate_state`: result of
        `hidden_gate_state(net, pgate)` for exactly this `net` (ignored when it does not match)."""
        E, H, W, _ = net.shape
        if not self.gate_state_matches(gate_state, net, pgate):
            gate_state = None
        if gate_state is None:
            self._gate_gen += 1  # this call writes the global-context scratch (extra, glo) a staged state of another graph reads
        c1 = self._buf("c1", (E, H, W, 128))
        f1 = self._buf("f1", (E, H, W, 128))
        zb = self._buf("z", (E, H, W, 128))
        rnet = self._buf("rnet", (E, H, W, 128))
        hbuf = self._buf("h", (E, H, W, 384))
        dw = self._buf("dw", (E, H, W, 4), torch.float32)
        glo = self._buf("glo", (E, 128), torch.float32)
        if net_out is None:
            net_out = torch.empty_like(net)
        if native:
            return self._forward_native(net, xbuf, corr, motn, ix, n_src, net_out, want_upmask, csr, pgate,
                                        (c1, f1, zb, rnet, hbuf, dw, glo), gate_state)
        # encoders (droid_net.py:481-482).  `corr` may be a deferred lookup ("lookup", levels, coords): lookup and the
        # first 1x1 convolution then run as ONE kernel and the [E,h,w,200] tensor never exists
        if isinstance(corr, tuple):
            from ..ext import droid_net_ext
            droid_net_ext.corr_lookup_conv1x1(corr[1], corr[2], self.corr0.packed, self.corr0.bias, c1, act="relu",
                                              slots=corr[3] if len(corr) > 3 else None,
                                              grid=corr[4] if len(corr) > 4 else None)
        else:
            self._conv(self.corr0, corr, 0, E, H, W, y=c1, act="relu", cin=CORR_CH)
        self._conv(self.corr2, c1, 0, E, H, W, y=xbuf, y_coff=128, act="relu")
        self._conv(self.flow0, motn, 0, E, H, W, y=f1, act="relu")
        self._conv(self.flow2, f1, 0, E, H, W, y=xbuf, y_coff=256, act="relu")
        # global context (droid_net.py:392-393) and its three 1x1s (a [E,128] x [128,384] product)
        extra = self._buf("extra", (E, 384), torch.float32)
        if gate_state is None:
            glo.zero_()
            self._conv(self.gw, net, 0, E, H, W, mode="glo", net=net, fout=glo)
            check(lib().vipe_glo_context(ptr(glo), ptr(self.glo_w), ptr(self.glo_b), ptr(extra), E, H * W, stream_ptr(glo)),
                  "glo_context")  # [E,128] x [128,384] + bias, / HW
        # gates (droid_net.py:395-399)
        if gate_state is not None:  # hidden-state part precomputed as well (hidden_gate_state): 192 channels left for z|r
            Es = gate_state["n_staged"]
            self._conv(self.zr_x, xbuf, 128, Es, H, W, y=zb, y2=rnet, net=net, mode="zr", extra=extra,
                       accinit=gate_state["pzr"], ai_coff=0, cin=192)
            if Es < E:  # the edges whose z|r part was not staged: the unsplit convolution on their slice
                self._conv(self.zr_s, net[Es:], 0, E - Es, H, W, x1=xbuf[Es:], x1_coff=128, split=128, y=zb[Es:], y2=rnet[Es:],
                           net=net[Es:], mode="zr", extra=extra[Es:], accinit=pgate[Es:], ai_coff=0)
            self._conv(self.q_s, rnet, 0, E, H, W, x1=xbuf, x1_coff=128, split=128, y=net_out, net=net, z=zb, mode="q",
                       extra=extra, extra_off=256, accinit=pgate, ai_coff=256)
        elif pgate is not None:  # context part precomputed (gate_context): 320 input channels, accumulators start at it
            self._conv(self.zr_s, net, 0, E, H, W, x1=xbuf, x1_coff=128, split=128, y=zb, y2=rnet, net=net, mode="zr",
                       extra=extra, accinit=pgate, ai_coff=0)
            self._conv(self.q_s, rnet, 0, E, H, W, x1=xbuf, x1_coff=128, split=128, y=net_out, net=net, z=zb, mode="q",
                       extra=extra, extra_off=256, accinit=pgate, ai_coff=256)
        else:
            self._conv(self.zr, net, 0, E, H, W, x1=xbuf, split=128, y=zb, y2=rnet, net=net, mode="zr", extra=extra)
            self._conv(self.q, rnet, 0, E, H, W, x1=xbuf, split=128, y=net_out, net=net, z=zb, mode="q", extra=extra,
                       extra_off=256)
        # heads + first aggregation conv on net' (droid_net.py:486-487, 418)
        self._conv(self.heads0, net_out, 0, E, H, W, y=hbuf, act="relu")
        self._conv(self.heads2, hbuf, 0, E, H, W, mode="heads", fout=dw, cin=256)
        eta = upmask = None
        if ix is not None:
            if n_src is None:  # the reference syncs here too (scatter.py:40: int(index.max()) + 1)
                n_src = int(ix.max().item()) + 1 if ix.numel() else 0
            # scatter_mean over the edges of each source node (droid_net.py:420-421): deterministic segmented mean
            # over a CSR of the edges (built once per edge set by the caller, or here)
            if csr is None:
                csr = segment_csr(ix, n_src)
            order, rowptr = csr
            agg = self._buf("agg", (n_src, H, W, 128))
            check(lib().vipe_segment_mean_nhwc_f16(ptr(hbuf), 384, 256, ptr(order), ptr(rowptr), ptr(agg), n_src, H * W,
                                                   128, stream_ptr(hbuf)), "segment_mean")
            a2 = self._buf("a2", (n_src, H, W, 128))
            eta = self._buf("eta", (n_src, H, W), torch.float32)
            self._conv(self.agg2, agg, 0, n_src, H, W, y=a2, act="relu")
            self._conv(self.eta, a2, 0, n_src, H, W, mode="eta", fout=eta)
            if want_upmask:
                upmask = torch.empty((n_src, H, W, 576), dtype=torch.float16, device=self.device)
                self._conv(self.upmask, a2, 0, n_src, H, W, y=upmask)
        return net_out, dw, eta, upmask

    def _forward_native(self, net, xbuf, corr, motn, ix, n_src, net_out, want_upmask, csr, pgate, scratch,
                        gate_state=None):
        c1, f1, zb, rnet, hbuf, dw, glo = scratch
        E, H, W, _ = net.shape
        extra = self._buf("extra", (E, 384), torch.float32)
        eta = agg = a2 = order = rowptr = None
        if ix is not None:
            if n_src is None:  # the reference syncs here too (scatter.py:40: int(index.max()) + 1)
                n_src = int(ix.max().item()) + 1 if ix.numel() else 0
            if csr is None:
                csr = segment_csr(ix, n_src)
            order, rowptr = csr
            agg = self._buf("agg", (n_src, H, W, 128))
            a2 = self._buf("a2", (n_src, H, W, 128))
            eta = self._buf("eta", (n_src, H, W), torch.float32)  # persistent: callers consume it before the next call
        lookup = isinstance(corr, tuple)
        tensors = (net, net_out, xbuf, motn, pgate, c1, f1, zb, rnet, hbuf, dw, glo, extra, order, rowptr, agg, a2, eta) + \
            ((tuple(corr[1]) + (corr[2],) + ((corr[3],) if len(corr) > 3 and corr[3] is not None else ())) if lookup else (corr,))
        pzr = gate_state["pzr"] if gate_state is not None else None
        two_streams = E >= self.op_side_min_edges
        if two_streams and self._op_side is None:
            self._op_side = torch.cuda.Stream(device=self.device)
        key = tuple(0 if t is None else t.data_ptr() for t in tensors) + (E, H, W, n_src or 0, two_streams,
                                                                         0 if pzr is None else pzr.data_ptr(),
                                                                         0 if pzr is None else int(gate_state["n_staged"]))
        b = self._bdesc.get(key)
        if b is None:
            if len(self._bdesc) > 16:
                self._bdesc.clear()
            b = UpdateBuffers()
            b.E, b.H, b.W, b.n_src = E, H, W, int(n_src or 0) if ix is not None else 0
            if lookup:
                lv = corr[1]
                for i in range(4):
                    b.levels[i] = lv[i].data_ptr()
                b.coords = corr[2].data_ptr()
                b.slots = corr[3].data_ptr() if len(corr) > 3 and corr[3] is not None else None
                if len(corr) > 4 and corr[4] is not None:  # (h, w) of the targets: a padded blocked store does not show them
                    b.h2, b.w2 = int(corr[4][0]), int(corr[4][1])
                else:
                    b.h2, b.w2 = int(lv[2].shape[3]) << 2, int(lv[2].shape[4]) << 2
                b.pyramid_layout = 1 if lv[0].dim() == 7 else 0
            else:
                b.corr = corr.data_ptr()
            for name, t in dict(motn=motn, net=net, net_out=net_out, xbuf=xbuf, pgate=pgate, c1=c1, f1=f1, zb=zb, rnet=rnet,
                                hbuf=hbuf, dw=dw, glo=glo, extra=extra, order=order, rowptr=rowptr, agg=agg, a2=a2,
                                eta=eta).items():
                setattr(b, name, None if t is None else t.data_ptr())
            if pzr is not None:
                b.pzr, b.gate_state = pzr.data_ptr(), int(gate_state["n_staged"])
            if two_streams:
                b.side_stream = self._op_side.cuda_stream
            self._bdesc[key] = b
        check(lib().vipe_update_operator(ctypes.addressof(self._wdesc), ctypes.addressof(b), stream_ptr(net)),
              "update_operator")
        upmask = None
        if want_upmask and ix is not None:
            upmask = torch.empty((n_src, H, W, 576), dtype=torch.float16, device=self.device)
            self._conv(self.upmask, a2, 0, n_src, H, W, y=upmask)
        return net_out, dw, eta, upmask

    # ------------------------------------------------------------------ reference-shaped entry point
    @torch.no_grad()
    def forward(self, net, inp, corr, flow=None, ix=None, skip_upmask=False, n_src=None):
        """Reference signature and return structure (droid_net.py:467-499): NCHW [1,E,C,h,w] in and out."""
        batch, num, ch, ht, wd = net.shape
        E = batch * num
        dev = net.device
        f16 = torch.float16
        net_n = net.reshape(E, 128, ht, wd).to(f16).permute(0, 2, 3, 1).contiguous()
        xbuf = torch.empty((E, ht, wd, 320), dtype=f16, device=dev)
        xbuf[..., 0:128] = inp.reshape(E, 128, ht, wd).permute(0, 2, 3, 1)
        corr_n = torch.zeros((E, ht, wd, CORR_CH), dtype=f16, device=dev)
        corr_n[..., :196] = corr.reshape(E, 196, ht, wd).permute(0, 2, 3, 1)
        if flow is None:
            motn_n = torch.zeros((E, ht, wd, 4), dtype=f16, device=dev)
        else:
            motn_n = flow.reshape(E, 4, ht, wd).to(f16).permute(0, 2, 3, 1).contiguous()
        ixd = ix.to(dev) if ix is not None else None
        pgate = self.gate_context(xbuf) if self.supports_gate_split(ht, wd) else None
        net_o, dw, eta, upmask = self.forward_nhwc(net_n, xbuf, corr_n, motn_n, ixd, n_src, want_upmask=not skip_upmask,
                                                   pgate=pgate)
        net_out = net_o.permute(0, 3, 1, 2).reshape(batch, num, 128, ht, wd)
        dwv = dw.view(batch, num, ht, wd, 4).to(f16)
        delta, weight = dwv[..., 0:2].contiguous(), dwv[..., 2:4].contiguous()
        if ix is None:
            return net_out, delta, weight
        n_src = eta.shape[0]
        if upmask is not None:
            upmask = upmask.permute(0, 3, 1, 2).reshape(batch, n_src, 576, ht, wd)
        return net_out, delta, weight, eta.clone().view(batch, n_src, ht, wd), upmask
