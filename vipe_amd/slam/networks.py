"""Flow-update operator and correlation blocks - host-side mirror of vipe/slam/networks/droid_net.py.

Same class names, constructor arguments, call signatures and state-dict keys as the reference
(CorrBlock droid_net.py:48-102, AltCorrBlock :121-176, ConvGRU :373-400, GraphAgg :403-429,
UpdateModule :432-499) so `FactorGraph.update` host code reads unchanged.  The arithmetic runs in the
HIP library (vipe_amd/csrc) through `vipe_amd.ext.droid_net_ext`; parameters live in ordinary
nn.Conv2d holders so reference checkpoints (`update.*` keys) load with `load_state_dict`.
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from .._lib import upload
from ..ext import droid_net_ext
from ..ext.scatter import scatter_mean


class CorrBlock:
    """All-pairs correlation pyramid + 7x7 bilinear lookup (droid_net.py:48-102).

    Layout: level i is [E, h1, w1, h2/2^i, w2/2^i], contiguous, dtype of the feature maps (fp16 under
    autocast in the reference, factor_graph.py:119) - 25.1 MB/edge at 48x64.  `from_buffer` builds the same pyramid
    from frame indices in the BLOCKED internal layout (include/vipe_amd.h) that the fused lookup kernel reads with half
    the memory sectors; `corr_pyramid` then converts on access.  Grids that are not multiples of 8 x 64 (41 x 73 for 16:9
    video) are always held blocked - their reference layout has no 16-byte aligned rows.
    """

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3):
        self.num_levels = num_levels
        self.radius = radius
        batch, num, dim, ht, wd = fmap1.shape
        self.ht, self.wd = ht, wd
        f1, f2 = fmap1.reshape(batch * num, dim, ht, wd), fmap2.reshape(batch * num, dim, ht, wd)
        blocked = (f1.is_cuda and f1.dtype == f2.dtype and droid_net_ext.fused_build_covers(dim, ht, wd, num_levels, f1.dtype)
                   and not droid_net_ext.blocked_dims(ht, wd)[5])
        self.levels = droid_net_ext.corr_pyramid_build(f1, f2, num_levels,
                                                       droid_net_ext.BLOCKED if blocked else droid_net_ext.REFERENCE)

    @classmethod
    def from_buffer(cls, fmaps, idx1, idx2, num_levels=4, radius=3, frame_range=None):
        """fmaps [n_frames,C,h,w]; edge e correlates frame idx1[e] with frame idx2[e] - the gathered copies
        `fmaps[idx1]`, `fmaps[idx2]` (factor_graph.py:147-148) are never made when the fused kernel covers the shape.
        `frame_range` (lo, hi): see droid_net_ext.corr_pyramid_build_indexed."""
        n, C, ht, wd = fmaps.shape
        if not (fmaps.is_cuda and droid_net_ext.fused_build_covers(C, ht, wd, num_levels, fmaps.dtype)):
            return cls(fmaps[idx1][None], fmaps[idx2][None], num_levels, radius)
        self = cls.__new__(cls)
        self.num_levels, self.radius, self.ht, self.wd = num_levels, radius, ht, wd
        self.levels = droid_net_ext.corr_pyramid_build_indexed(fmaps, idx1.contiguous(), idx2.contiguous(),
                                                               num_levels=num_levels, frame_range=frame_range)
        return self

    def _fused_lookup_covers(self, lv):
        """4 levels, radius 3, fp16 on the device; the reference layout needs 8-element row chunks down to level 3"""
        return (self.num_levels == 4 and self.radius == 3 and lv[0].dtype == torch.float16 and lv[0].is_cuda
                and (self.wd >> 3) >= 1 and (self.ht >> 3) >= 1 and (lv[0].dim() == 7 or (self.wd >> 3) % 8 == 0))

    @property
    def corr_pyramid(self):
        """the reference's attribute: list of [E,h,w,h>>i,w>>i] (a converted copy when the store is blocked)"""
        return droid_net_ext.pyramid_to_reference(self.levels, self.ht, self.wd)

    @corr_pyramid.setter
    def corr_pyramid(self, levels):
        self.levels = list(levels)

    def __call__(self, coords):
        batch, num, ht, wd, _ = coords.shape
        out = droid_net_ext.corr_pyramid_lookup(self.corr_pyramid, coords.reshape(batch * num, ht, wd, 2), self.radius)
        return out.view(batch, num, -1, ht, wd)

    def lookup_nhwc(self, coords, channel_stride=200):
        """coords [E,h,w,2] -> [E,h,w,channel_stride] fp16 channels-last (the GRU's input layout)."""
        return droid_net_ext.corr_pyramid_lookup_nhwc(self.corr_pyramid, coords, self.radius, channel_stride)

    def lookup_deferred(self, coords):
        """Handle for the fused lookup + correlation-encoder kernel (`UpdateEngine.forward_nhwc` consumes it), or the
        materialised channels-last lookup when the fused kernel does not cover this pyramid."""
        lv = self.levels
        if self._fused_lookup_covers(lv):
            return ("lookup", lv, coords.contiguous(), None, (self.ht, self.wd))
        return self.lookup_nhwc(coords)

    def cat(self, other):
        assert all(a.dim() == b.dim() for a, b in zip(self.levels, other.levels)), "layouts differ"
        self.levels = [torch.cat([a, b], 0) for a, b in zip(self.levels, other.levels)]
        return self

    def __getitem__(self, index):
        self.levels = [lv[index] for lv in self.levels]
        return self

    @staticmethod
    def corr(fmap1, fmap2):
        """all-pairs correlation (droid_net.py:94-102): (fmap1/4)^T (fmap2/4) -> [B,num,ht,wd,ht,wd]"""
        batch, num, dim, ht, wd = fmap1.shape
        vol = droid_net_ext.corr_volume(fmap1.reshape(batch * num, dim, ht, wd), fmap2.reshape(batch * num, dim, ht, wd))
        return vol.view(batch, num, ht, wd, ht, wd)


def _to_blocked(levels, ht, wd):
    """reference-layout levels -> BLOCKED (inverse of droid_net_ext.pyramid_to_reference)"""
    return droid_net_ext.pyramid_to_blocked(levels, ht, wd)


class CorrPool:
    """Edge-indexed store of correlation pyramids for a graph whose edges come and go every keyframe
    (factor_graph.py:147-152 appends with `corr.cat`, :194-196 drops with `corr[~mask]` - each a copy of the whole
    pyramid, 25 MB per edge).  Here the level buffers have spare capacity and edge e owns slot `slots[e]`: `add_edges`
    has the build kernel write the new edges' pyramids straight into free slots (from frame indices - neither the
    gathered feature maps nor a temporary pyramid exist), removing only edits the slot vector, and the fused lookup
    kernel follows the indirection.  Being private, the store uses the BLOCKED layout (include/vipe_amd.h) whenever
    the fused kernels cover the shape; `corr_pyramid` materialises the reference layout.  Same call surface as
    CorrBlock (`cat`, `__getitem__`, `__call__`, `lookup_nhwc`, `lookup_deferred`, `corr_pyramid`)."""

    def __init__(self, num_levels=4, radius=3, capacity=64):
        self.num_levels, self.radius, self.capacity = num_levels, radius, capacity
        self.pool = None
        self.blocked = False
        self.ht = self.wd = None
        self.slots = None        # int32 [E] on the device
        self._slots_host = []    # the same, host side (edge bookkeeping never reads the device copy back)
        self._free = []

    def __len__(self):
        return len(self._slots_host)

    def _reserve(self, need, shapes_of, dtype, device):
        """make `need` free slots available; shapes_of(cap) -> per-level shapes of a `cap`-slot store"""
        if self.pool is None:
            cap = self.capacity
            while cap < need:
                cap *= 2
            self.pool = [torch.empty(s, dtype=dtype, device=device) for s in shapes_of(cap)]
            self._free = list(range(cap))
            return
        cap = new_cap = self.pool[0].shape[0]
        while len(self._free) + (new_cap - cap) < need:
            new_cap *= 2
        if new_cap > cap:
            self.pool = [torch.cat([p, torch.empty((new_cap - cap,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)], 0)
                         for p in self.pool]
            self._free += list(range(cap, new_cap))

    def _take(self, k, device):
        ids = [self._free.pop(0) for _ in range(k)]
        self._slots_host += ids
        new = upload(ids, device, torch.int32)
        self.slots = new if self.slots is None else torch.cat([self.slots, new], 0)
        return ids, new

    def add_edges(self, fmaps, idx1, idx2, frame_range=None):
        """append the edges (frame idx1[e] -> frame idx2[e]) of fmaps [n_frames,C,h,w]; `frame_range` (lo, hi): the
        frames the indices lie in (droid_net_ext.corr_pyramid_build_indexed)"""
        n, C, ht, wd = fmaps.shape
        k = int(idx1.shape[0])
        fused = fmaps.is_cuda and droid_net_ext.fused_build_covers(C, ht, wd, self.num_levels, fmaps.dtype)
        if not fused or (self.pool is not None and not self.blocked):
            return self.cat(CorrBlock(fmaps[idx1][None], fmaps[idx2][None], self.num_levels, self.radius))
        self.blocked, self.ht, self.wd = True, ht, wd
        self._reserve(k, lambda cap: droid_net_ext.pyramid_level_shapes(cap, ht, wd, self.num_levels, droid_net_ext.BLOCKED),
                      fmaps.dtype, fmaps.device)
        _, new = self._take(k, fmaps.device)
        droid_net_ext.corr_pyramid_build_indexed(fmaps, idx1.contiguous(), idx2.contiguous(), levels=self.pool, slots=new,
                                                 num_levels=self.num_levels, frame_range=frame_range)
        return self

    def cat(self, other):
        """append the edges of a CorrBlock (its volumes are copied into free slots)"""
        lv = other.levels if hasattr(other, "levels") else other.corr_pyramid
        k = lv[0].shape[0]
        if self.ht is None:
            self.ht, self.wd = (other.ht, other.wd) if hasattr(other, "ht") else (int(x) for x in lv[-1].shape[1:3])
        if self.pool is not None and self.blocked and lv[0].dim() != 7:
            lv = _to_blocked(lv, self.ht, self.wd)
        elif self.pool is None:
            self.blocked = lv[0].dim() == 7
        self._reserve(k, lambda cap: [(cap,) + tuple(l.shape[1:]) for l in lv], lv[0].dtype, lv[0].device)
        ids, _ = self._take(k, lv[0].device)
        idt = upload(ids, lv[0].device)
        for p, l in zip(self.pool, lv):
            p.index_copy_(0, idt, l)
        return self

    def __getitem__(self, index):
        """keep the edges at positions `index` (1-D integer array / tensor or boolean mask), in that order"""
        import numpy as np
        idx = index.detach().cpu().numpy() if torch.is_tensor(index) else np.asarray(index)
        if idx.dtype == np.bool_:
            idx = np.flatnonzero(idx)
        kept = [self._slots_host[int(i)] for i in idx]
        gone = set(self._slots_host) - set(kept)
        self._free += sorted(gone)
        self._slots_host = kept
        self.slots = upload(kept, self.pool[0].device, torch.int32)
        return self

    @property
    def corr_pyramid(self):
        """materialised [E, h, w, h>>i, w>>i] levels in edge order (reference layout; copies)"""
        idx = self.slots.long()
        lv = [p.index_select(0, idx) for p in self.pool]
        return droid_net_ext.pyramid_to_reference(lv, self.ht, self.wd) if self.blocked else lv

    def __call__(self, coords):
        batch, num, ht, wd, _ = coords.shape
        out = droid_net_ext.corr_pyramid_lookup(self.corr_pyramid, coords.reshape(batch * num, ht, wd, 2), self.radius)
        return out.view(batch, num, -1, ht, wd)

    def lookup_nhwc(self, coords, channel_stride=200):
        return droid_net_ext.corr_pyramid_lookup_nhwc(self.corr_pyramid, coords, self.radius, channel_stride)

    def lookup_deferred(self, coords):
        lv = self.pool
        if CorrBlock._fused_lookup_covers(self, lv):
            return ("lookup", lv, coords.contiguous(), self.slots, (self.ht, self.wd))
        return self.lookup_nhwc(coords)


class AltCorrBlock:
    """Volume-free correlation (droid_net.py:121-176): pyramid of channels-last fmaps/4, looked up on the fly."""

    def __init__(self, fmaps, num_levels=4, radius=3):
        self.num_levels = num_levels
        self.radius = radius
        B, N, C, H, W = fmaps.shape
        f = fmaps.view(B * N, C, H, W) / 4.0
        self.pyramid = []
        for i in range(num_levels):
            self.pyramid.append(f.permute(0, 2, 3, 1).contiguous().view(B, N, H // 2**i, W // 2**i, C))
            if i + 1 < num_levels:
                f = F.avg_pool2d(f, 2, stride=2)

    def corr_fn(self, coords, ii, jj):
        B, N, H, W, S, _ = coords.shape
        coords = coords.permute(0, 1, 4, 2, 3, 5)
        corr_list = []
        for i in range(self.num_levels):
            fmap1_i = self.pyramid[0][:, ii]
            fmap2_i = self.pyramid[i][:, jj]
            coords_i = (coords / 2**i).reshape(B * N, S, H, W, 2).contiguous()
            fmap1_i = fmap1_i.reshape((B * N,) + fmap1_i.shape[2:])
            fmap2_i = fmap2_i.reshape((B * N,) + fmap2_i.shape[2:])
            (corr,) = droid_net_ext.altcorr_forward(fmap1_i.float(), fmap2_i.float(), coords_i, self.radius)
            corr_list.append(corr.view(B, N, S, -1, H, W).permute(0, 1, 3, 4, 5, 2))
        return torch.cat(corr_list, dim=2)

    def __call__(self, coords, ii, jj):
        squeeze = coords.dim() == 5
        if squeeze:
            coords = coords.unsqueeze(-2)
        corr = self.corr_fn(coords, ii, jj)
        if squeeze:
            corr = corr.squeeze(-1)
        return corr.contiguous()


class ConvGRU(nn.Module):
    """Parameter holder with the reference key layout (droid_net.py:373-385)."""

    def __init__(self, h_planes=128, i_planes=128):
        super().__init__()
        self.convz = nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1)
        self.convr = nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1)
        self.convq = nn.Conv2d(h_planes + i_planes, h_planes, 3, padding=1)
        self.w = nn.Conv2d(h_planes, h_planes, 1, padding=0)
        self.convz_glo = nn.Conv2d(h_planes, h_planes, 1, padding=0)
        self.convr_glo = nn.Conv2d(h_planes, h_planes, 1, padding=0)
        self.convq_glo = nn.Conv2d(h_planes, h_planes, 1, padding=0)


class GraphAgg(nn.Module):
    """Parameter holder (droid_net.py:403-412)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(128, 128, 3, padding=1)
        self.conv2 = nn.Conv2d(128, 128, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.eta = nn.Sequential(nn.Conv2d(128, 1, 3, padding=1), nn.Softplus())
        self.upmask = nn.Sequential(nn.Conv2d(128, 8 * 8 * 9, 1, padding=0))


class UpdateModule(nn.Module):
    """RaftSLAM update operator (droid_net.py:432-499)."""

    def __init__(self):
        super().__init__()
        cor_planes = 4 * (2 * 3 + 1) ** 2
        self.corr_encoder = nn.Sequential(
            nn.Conv2d(cor_planes, 128, 1, padding=0), nn.ReLU(inplace=True),
            nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True))
        self.flow_encoder = nn.Sequential(
            nn.Conv2d(4, 128, 7, padding=3), nn.ReLU(inplace=True),
            nn.Conv2d(128, 64, 3, padding=1), nn.ReLU(inplace=True))
        self.weight = nn.Sequential(
            nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(128, 2, 3, padding=1), nn.Sigmoid())
        self.delta = nn.Sequential(
            nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(128, 2, 3, padding=1))
        self.gru = ConvGRU(128, 128 + 128 + 64)
        self.agg = GraphAgg()
        self._engine = None

    def engine(self, device):
        from .update_engine import UpdateEngine

        if self._engine is None or self._engine.device != device:
            self._engine = UpdateEngine(self, device)
        return self._engine

    def forward(self, net, inp, corr, flow=None, ix=None, skip_upmask=False, n_src=None):
        """net, inp [1,E,128,h,w]; corr [1,E,196,h,w]; flow [1,E,4,h,w]; ix [E] -> source-node slot.

        Returns (net, delta[1,E,h,w,2], weight[1,E,h,w,2], eta[1,Nsrc,h,w], upmask) like the reference;
        `skip_upmask=True` returns None for upmask (the SLAM host code discards it, factor_graph.py:269).
        """
        return self.engine(net.device).forward(net, inp, corr, flow, ix, skip_upmask, n_src)
