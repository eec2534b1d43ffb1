"""Correlation: all-pairs volume + pyramid, windowed bilinear lookup, on-the-fly (alt) correlation.

ORACLE (test infrastructure). Follows
  vipe/slam/networks/droid_net.py:56-69,94-102   CorrBlock volume + avg-pool pyramid
  csrc/droid_net_ext/correlation_kernels.cu:22-66 corr_index_forward_kernel
  vipe/slam/networks/droid_net.py:130-163         AltCorrBlock pyramid / call
  csrc/droid_net_ext/altcorr_kernel.cu:26-138     altcorr_forward_kernel
The two kernels have no CPU path in the reference: restated from the kernel text
("parity unpinned" by any reference output), cross-checked in tests against
F.grid_sample.  Half precision follows c10::Half arithmetic: every `*` and `+=`
is computed in float and rounded to half (c10/util/Half-inl.h operators), and
the bilinear weight is rounded to half before the multiply (kernel lines 56-62).
"""

import numpy as np
import torch
import torch.nn.functional as F


def corr_volume(fmap1, fmap2):
    """droid_net.py:94-102. fmap [B,num,C,ht,wd] torch -> [B,num,ht,wd,ht,wd]."""
    b, n, c, ht, wd = fmap1.shape
    f1 = fmap1.reshape(b * n, c, ht * wd) / 4.0
    f2 = fmap2.reshape(b * n, c, ht * wd) / 4.0
    return torch.matmul(f1.transpose(1, 2), f2).view(b, n, ht, wd, ht, wd)


def corr_pyramid(fmap1, fmap2, num_levels=4):
    """droid_net.py:56-69 -> list of [B*num,h1,w1,h2/2^i,w2/2^i]."""
    corr = corr_volume(fmap1, fmap2)
    b, n, h1, w1, h2, w2 = corr.shape
    corr = corr.reshape(b * n * h1 * w1, 1, h2, w2)
    pyr = []
    for i in range(num_levels):
        pyr.append(corr.view(b * n, h1, w1, h2 // 2**i, w2 // 2**i))
        if i + 1 < num_levels:
            corr = F.avg_pool2d(corr, 2, stride=2)
    return pyr


def corr_index_forward(volume, coords, radius):
    """correlation_kernels.cu:22-66. volume [B,h1,w1,h2,w2] (f16/f32/f64 numpy), coords [B,2,h1,w1] f32.

    Returns corr [B,2r+1,2r+1,h1,w1] in volume's dtype; output index [i][j]: i <-> x offset, j <-> y offset.
    Accumulation order per output follows the kernel's i-outer / j-inner tap loop.
    """
    vol = np.asarray(volume)
    T = vol.dtype
    B, h1, w1, h2, w2 = vol.shape
    r = radius
    rd = 2 * r + 1
    x0 = np.asarray(coords[:, 0], dtype=np.float32)
    y0 = np.asarray(coords[:, 1], dtype=np.float32)
    fx, fy = np.floor(x0), np.floor(y0)
    dx = (x0 - fx).astype(np.float32)
    dy = (y0 - fy).astype(np.float32)
    one = np.float32(1.0)
    w_nw = (dx * dy).astype(T)
    w_ne = (dx * (one - dy)).astype(T)
    w_sw = ((one - dx) * dy).astype(T)
    w_se = ((one - dx) * (one - dy)).astype(T)
    acc = T if T != np.float16 else np.float32  # c10::Half ops: compute in float, round to half
    out = np.zeros((B, rd, rd, h1, w1), dtype=T)
    ix = fx.astype(np.int64)
    iy = fy.astype(np.int64)
    bb, yy, xx = np.meshgrid(np.arange(B), np.arange(h1), np.arange(w1), indexing="ij")

    def madd(o, s, w):
        prod = (s.astype(acc) * w.astype(acc)).astype(T)
        return (o.astype(acc) + prod.astype(acc)).astype(T)

    with np.errstate(invalid="ignore", over="ignore"):
        for i in range(rd + 1):
            for j in range(rd + 1):
                x1 = ix - r + i
                y1 = iy - r + j
                inb = (y1 >= 0) & (y1 < h2) & (x1 >= 0) & (x1 < w2)
                s = vol[bb, yy, xx, np.clip(y1, 0, h2 - 1), np.clip(x1, 0, w2 - 1)]
                if i > 0 and j > 0:
                    out[:, i - 1, j - 1] = np.where(inb, madd(out[:, i - 1, j - 1], s, w_nw), out[:, i - 1, j - 1])
                if i > 0 and j < rd:
                    out[:, i - 1, j] = np.where(inb, madd(out[:, i - 1, j], s, w_ne), out[:, i - 1, j])
                if i < rd and j > 0:
                    out[:, i, j - 1] = np.where(inb, madd(out[:, i, j - 1], s, w_sw), out[:, i, j - 1])
                if i < rd and j < rd:
                    out[:, i, j] = np.where(inb, madd(out[:, i, j], s, w_se), out[:, i, j])
    return out


def corr_index_backward(volume_shape, coords, corr_grad, radius, dtype=np.float32):
    """correlation_kernels.cu:68-113 (adjoint)."""
    B, h1, w1, h2, w2 = volume_shape
    r = radius
    rd = 2 * r + 1
    g = np.asarray(corr_grad, dtype=dtype)
    x0 = np.asarray(coords[:, 0], dtype=np.float32)
    y0 = np.asarray(coords[:, 1], dtype=np.float32)
    fx, fy = np.floor(x0), np.floor(y0)
    dx = (x0 - fx)
    dy = (y0 - fy)
    ix, iy = fx.astype(np.int64), fy.astype(np.int64)
    out = np.zeros(volume_shape, dtype=dtype)
    bb, yy, xx = np.meshgrid(np.arange(B), np.arange(h1), np.arange(w1), indexing="ij")
    for i in range(rd + 1):
        for j in range(rd + 1):
            x1 = ix - r + i
            y1 = iy - r + j
            inb = (y1 >= 0) & (y1 < h2) & (x1 >= 0) & (x1 < w2)
            acc = np.zeros((B, h1, w1), dtype=dtype)
            if i > 0 and j > 0:
                acc += g[:, i - 1, j - 1] * (dx * dy).astype(dtype)
            if i > 0 and j < rd:
                acc += g[:, i - 1, j] * (dx * (1 - dy)).astype(dtype)
            if i < rd and j > 0:
                acc += g[:, i, j - 1] * ((1 - dx) * dy).astype(dtype)
            if i < rd and j < rd:
                acc += g[:, i, j] * ((1 - dx) * (1 - dy)).astype(dtype)
            np.add.at(out, (bb[inb], yy[inb], xx[inb], y1[inb], x1[inb]), acc[inb])
    return out


def corr_lookup(pyramid, coords, radius=3):
    """CorrBlock.__call__, droid_net.py:71-82. pyramid: list of numpy levels; coords [1,E,ht,wd,2] numpy f32.

    Returns [1,E,4*(2r+1)^2,ht,wd]."""
    b, n, ht, wd, _ = coords.shape
    c = np.ascontiguousarray(np.transpose(coords, (0, 1, 4, 2, 3))).reshape(b * n, 2, ht, wd)
    outs = []
    for i, lvl in enumerate(pyramid):
        ci = (c / np.float32(2**i)).astype(np.float32)
        o = corr_index_forward(np.asarray(lvl), ci, radius)
        outs.append(o.reshape(b, n, -1, ht, wd))
    return np.concatenate(outs, axis=2)


def alt_pyramid(fmaps, num_levels=4):
    """AltCorrBlock.__init__, droid_net.py:130-142. fmaps [B,N,C,H,W] torch -> list of [B,N,H/2^i,W/2^i,C]."""
    B, N, C, H, W = fmaps.shape
    f = fmaps.reshape(B * N, C, H, W) / 4.0
    pyr = []
    for i in range(num_levels):
        pyr.append(f.permute(0, 2, 3, 1).contiguous().view(B, N, H // 2**i, W // 2**i, C))
        if i + 1 < num_levels:
            f = F.avg_pool2d(f, 2, stride=2)
    return pyr


def altcorr_forward(fmap1, fmap2, coords, radius):
    """altcorr_kernel.cu:26-138. fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,N,H1,W1,2] (numpy f32).

    Returns corr [B,N,(2r+1)^2,H1,W1]; channel = iy + rd*ix (x-offset major), i.e. the same order as
    corr_index_forward.  The kernel accumulates the C-channel dot product in slabs of 32 channels, each
    slab's partial dot product being splatted separately (lines 50, 95-133); float32 only here.
    """
    f1 = np.asarray(fmap1, dtype=np.float32)
    f2 = np.asarray(fmap2, dtype=np.float32)
    co = np.asarray(coords, dtype=np.float32)
    B, H1, W1, C = f1.shape
    _, H2, W2, _ = f2.shape
    N = co.shape[1]
    r = radius
    rd = 2 * r + 1
    out = np.zeros((B, N, rd * rd, H1, W1), dtype=np.float32)
    one = np.float32(1.0)
    bb = np.arange(B)[:, None, None]
    for c0 in range(0, C, 32):
        a = f1[..., c0:c0 + 32]
        for n in range(N):
            x = co[:, n, :, :, 0]
            y = co[:, n, :, :, 1]
            fx, fy = np.floor(x), np.floor(y)
            dx, dy = x - fx, y - fy
            ix, iy_ = fx.astype(np.int64), fy.astype(np.int64)
            for iy in range(rd + 1):
                for ixx in range(rd + 1):
                    h2 = iy_ - r + iy
                    w2 = ix - r + ixx
                    inb = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)
                    g = f2[bb, np.clip(h2, 0, H2 - 1), np.clip(w2, 0, W2 - 1), c0:c0 + 32]
                    g = np.where(inb[..., None], g, np.float32(0))
                    s = np.zeros((B, H1, W1), dtype=np.float32)
                    for k in range(a.shape[-1]):  # sequential float accumulation, kernel line 96
                        s = s + a[..., k] * g[..., k]
                    if iy > 0 and ixx > 0:
                        out[:, n, (iy - 1) + rd * (ixx - 1)] += s * (dy * dx)
                    if iy > 0 and ixx < rd:
                        out[:, n, (iy - 1) + rd * ixx] += s * (dy * (one - dx))
                    if iy < rd and ixx > 0:
                        out[:, n, iy + rd * (ixx - 1)] += s * ((one - dy) * dx)
                    if iy < rd and ixx < rd:
                        out[:, n, iy + rd * ixx] += s * ((one - dy) * (one - dx))
    return out
