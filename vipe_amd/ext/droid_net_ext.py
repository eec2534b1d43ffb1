"""droid_net_ext: correlation lookup ops (reference: csrc/droid_net_ext/droid.cpp:57-63)."""

import ctypes

import torch

from .._lib import DTYPE_CODE, check, check_gpu_contig, lib, ptr, require, stream_ptr


def corr_index_forward(volume, coords, radius):
    """droid.cpp:20-27 -> correlation_kernels.cu:115-136. volume [B,h1,w1,h2,w2]; coords [B,2,h1,w1] f32."""
    check_gpu_contig(volume, coords)
    require(volume.dim() == 5 and coords.dim() == 4 and coords.dtype == torch.float32, "bad corr_index_forward inputs")
    require(volume.dtype in DTYPE_CODE, "volume must be half/float/double")
    B, h1, w1, h2, w2 = volume.shape
    require(tuple(coords.shape) == (B, 2, h1, w1), "coords must be [B,2,h1,w1]")
    rd = 2 * radius + 1
    corr = torch.empty((B, rd, rd, h1, w1), dtype=volume.dtype, device=volume.device)
    check(lib().vipe_corr_index_forward(ptr(volume), ptr(coords), ptr(corr), B, h1, w1, h2, w2, radius,
                                        DTYPE_CODE[volume.dtype], stream_ptr(volume)), "corr_index_forward")
    return [corr]


def corr_index_backward(volume, coords, corr_grad, radius):
    """droid.cpp:29-37 -> correlation_kernels.cu:138-159."""
    check_gpu_contig(volume, coords, corr_grad)
    B, h1, w1, h2, w2 = volume.shape
    require(corr_grad.dtype == volume.dtype, "corr_grad dtype must match volume")
    grad = torch.empty_like(volume)
    check(lib().vipe_corr_index_backward(ptr(coords), ptr(corr_grad), ptr(grad), B, h1, w1, h2, w2, radius,
                                         DTYPE_CODE[volume.dtype], stream_ptr(volume)), "corr_index_backward")
    return [grad]


def altcorr_forward(fmap1, fmap2, coords, radius):
    """droid.cpp:39-45 -> altcorr_kernel.cu:266-290. fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,N,H1,W1,2]."""
    check_gpu_contig(fmap1, fmap2, coords)
    require(fmap1.dtype == fmap2.dtype and fmap1.dtype in (torch.float16, torch.float32), "fmaps must be half/float")
    require(coords.dtype == torch.float32 and coords.dim() == 5, "coords must be [B,N,H1,W1,2] float32")
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    rd = 2 * radius + 1
    corr = torch.empty((B, N, rd * rd, H1, W1), dtype=fmap1.dtype, device=fmap1.device)
    check(lib().vipe_altcorr_forward(ptr(fmap1), ptr(fmap2), ptr(coords), ptr(corr), B, H1, W1, H2, W2, N, C,
                                     radius, DTYPE_CODE[fmap1.dtype], stream_ptr(fmap1)), "altcorr_forward")
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    """droid.cpp:47-55 -> altcorr_kernel.cu:292-320 (float32 only)."""
    check_gpu_contig(fmap1, fmap2, coords, corr_grad)
    require(fmap1.dtype == torch.float32 and fmap2.dtype == torch.float32, "altcorr_backward is float32 only")
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    g1 = torch.zeros_like(fmap1)
    g2 = torch.zeros_like(fmap2)
    check(lib().vipe_altcorr_backward(ptr(fmap1), ptr(fmap2), ptr(coords), ptr(corr_grad), ptr(g1), ptr(g2), B, H1,
                                      W1, H2, W2, N, C, radius, stream_ptr(fmap1)), "altcorr_backward")
    return [g1, g2, torch.zeros_like(coords)]


# ------------------------------------------------------------------ fused entry points (MI355X design)


def corr_volume(fmap1, fmap2):
    """All-pairs correlation (fmap1/4)^T (fmap2/4), [E,C,h,w] x2 -> [E,h*w,h*w] (droid_net.py:94-102), dtype of the
    inputs: fp16 maps with 128 channels on the MFMA build kernel (level 0 of the pyramid), anything else on the
    library's plain tiled kernel (`vipe_corr_volume`)."""
    E, C, h, w = fmap1.shape
    if fmap1.is_cuda and fmap1.dtype == fmap2.dtype and fused_build_covers(C, h, w, 1, fmap1.dtype):
        return corr_pyramid_build(fmap1, fmap2, 1)[0].reshape(E, h * w, h * w)
    check_gpu_contig(fmap1, fmap2)
    require(fmap1.dtype == fmap2.dtype and fmap1.dtype in DTYPE_CODE, "fmaps must be half/float/double of one dtype")
    vol = torch.empty((E, h * w, h * w), dtype=fmap1.dtype, device=fmap1.device)
    check(lib().vipe_corr_volume(ptr(fmap1), ptr(fmap2), ptr(vol), E, C, h * w, DTYPE_CODE[fmap1.dtype], stream_ptr(fmap1)),
          "corr_volume")
    return vol


REFERENCE, BLOCKED = 0, 1  # VIPE_PYRAMID_* (include/vipe_amd.h)


def blocked_dims(h, w):
    """(G, S, R, level-2 pitch, level-3 pitch, direct) of the padded blocked layout (include/vipe_amd.h); `direct`: the build
    kernel tiles the grid straight from the [C][h*w] maps (w % 64 == 0, h % 8 == 0), no operand preparation"""
    d = (ctypes.c_int * 6)()
    check(lib().vipe_corr_blocked_dims(h, w, ctypes.cast(d, ctypes.c_void_p)), "corr_blocked_dims")
    return tuple(int(x) for x in d[:5]) + (bool(d[5]),)


def fused_build_covers(C, h, w, num_levels, dtype=torch.float16):
    """shapes of the fused volume + pyramid kernel: fp16 maps with 128 channels on any grid whose coarsest level is not
    empty (grids other than multiples of 8 x 64 go through `vipe_corr_prep`)"""
    return dtype == torch.float16 and C == 128 and num_levels <= 4 and (h >> (num_levels - 1)) > 0 and (w >> (num_levels - 1)) > 0


def pyramid_level_shapes(n, h, w, num_levels, layout):
    """per-level tensor shapes of `n` edges' pyramids in `layout`"""
    shapes = [(n, h, w, h >> i, w >> i) for i in range(num_levels)]
    if layout == BLOCKED:
        G, S, R, w2p, w3p, direct = blocked_dims(h, w)
        shapes[0] = (n, G, S * R, 64, 4, 4, 8)
        if num_levels > 1:
            shapes[1] = (n, G, S * R // 2, 64, 2, 4, 8)
        if not direct:  # padded grid: one slab per source pixel of every group of 64, padded rows
            if num_levels > 2:
                shapes[2] = (n, G * 64, R, w2p)
            if num_levels > 3:
                shapes[3] = (n, G * 64, R // 2, w3p)
    return shapes


def pyramid_to_reference(levels, h, w):
    """BLOCKED levels -> the reference's [n,h,w,h>>i,w>>i] tensors (copies; levels 2.. of an unpadded store are shared)"""
    out = list(levels)
    if not len(levels) or levels[0].dim() != 7:
        return out
    G, S, R, w2p, w3p, direct = blocked_dims(h, w)
    P = h * w
    for i in range(min(2, len(levels))):
        lv = levels[i]
        n, T, Ri = lv.shape[0], lv.shape[4], R >> i
        # [n, G, strip, rowgroup, p, tile, row, col] -> [n, G, p, rowgroup, row, strip, tile, col]
        x = lv.reshape(n, G, S, Ri, 64, T, 4, 8).permute(0, 1, 4, 3, 6, 2, 5, 7).reshape(n, G * 64, Ri * 4, S * T * 8)
        out[i] = x[:, :P, :h >> i, :w >> i].reshape(n, h, w, h >> i, w >> i).contiguous()
    if not direct:
        for i in range(2, len(levels)):
            out[i] = levels[i][:, :P, :h >> i, :w >> i].reshape(levels[i].shape[0], h, w, h >> i, w >> i).contiguous()
    return out


def pyramid_to_blocked(levels, h, w):
    """reference-layout levels [n,h,w,h>>i,w>>i] -> BLOCKED on the padded grid (inverse of pyramid_to_reference; zeros in
    the padding)"""
    G, S, R, w2p, w3p, direct = blocked_dims(h, w)
    P = h * w
    out = list(levels)
    n = levels[0].shape[0]
    for i in range(len(levels)):
        lv = levels[i].reshape(n, P, h >> i, w >> i)
        if i < 2:
            T, Ri = 4 >> i, R >> i
            x = lv.new_zeros((n, G * 64, Ri * 4, S * T * 8))
            x[:, :P, :h >> i, :w >> i] = lv
            x = x.reshape(n, G, 64, Ri, 4, S, T, 8).permute(0, 1, 5, 3, 2, 6, 4, 7)
            out[i] = x.reshape(n, G, S * Ri, 64, T, 4, 8).contiguous()
        elif not direct:
            x = lv.new_zeros((n, G * 64, R >> (i - 2), w2p if i == 2 else w3p))
            x[:, :P, :h >> i, :w >> i] = lv
            out[i] = x
    return out


def _level_ptrs(levels):
    return (ctypes.c_void_p * len(levels))(*[lv.data_ptr() for lv in levels])


def corr_prep(fmaps):
    """[fused] [n,128,h,w] f16 maps -> the zero-padded operand images of the general-grid build kernel (`vipe_corr_prep`)"""
    check_gpu_contig(fmaps)
    n, C, h, w = fmaps.shape
    per = int(lib().vipe_corr_prep_halves(C, h, w))
    require(per > 0 and fmaps.dtype == torch.float16, "corr_prep: fp16 maps with 128 channels")
    prep = torch.empty((n, per), dtype=torch.float16, device=fmaps.device)
    check(lib().vipe_corr_prep(ptr(fmaps), ptr(prep), n, C, h, w, stream_ptr(fmaps)), "corr_prep")
    return prep


def corr_pyramid_build(fmap1, fmap2, num_levels=4, layout=REFERENCE):
    """CorrBlock.__init__ (droid_net.py:56-69): list of [E,h,w,h>>i,w>>i] (`layout=BLOCKED`: the internal layout).

    fp16 feature maps with C = 128 go through the fused HIP kernel (volume + the three pooled levels in one pass) - for
    w % 64 == 0, h % 8 == 0 straight from the maps, for every other grid from their prepared operand images, blocked
    layout (converted back when the reference layout is asked for); other dtypes / channel counts take the library's
    plain kernels (`vipe_corr_volume`, `vipe_avg_pool2x2`)."""
    E, C, h, w = fmap1.shape
    if fmap1.dtype == fmap2.dtype and fmap1.is_cuda and fused_build_covers(C, h, w, num_levels, fmap1.dtype):
        check_gpu_contig(fmap1, fmap2)
        if blocked_dims(h, w)[5] and layout == REFERENCE:
            levels = [torch.empty((E, h, w, h >> i, w >> i), dtype=torch.float16, device=fmap1.device)
                      for i in range(num_levels)]
            check(lib().vipe_corr_pyramid_build(ptr(fmap1), ptr(fmap2), _level_ptrs(levels), E, C, h, w, num_levels,
                                                stream_ptr(fmap1)), "corr_pyramid_build")
            return levels
        idx = torch.arange(2 * E, device=fmap1.device)
        levels = corr_pyramid_build_indexed(torch.cat([fmap1, fmap2], 0), idx[:E], idx[E:], num_levels=num_levels,
                                            frame_range=(0, 2 * E))
        return levels if layout == BLOCKED else [lv.contiguous() for lv in pyramid_to_reference(levels, h, w)]
    require(layout == REFERENCE, "the blocked layout is the fp16 / 128-channel kernel's")
    check_gpu_contig(fmap1, fmap2)
    vol = corr_volume(fmap1, fmap2)
    levels = [vol.view(E, h, w, h, w)]
    hl, wl = h, w
    for i in range(1, num_levels):
        nxt = torch.empty((E, h, w, hl >> 1, wl >> 1), dtype=vol.dtype, device=vol.device)
        check(lib().vipe_avg_pool2x2(ptr(levels[-1]), ptr(nxt), E * h * w, hl, wl, DTYPE_CODE[vol.dtype], stream_ptr(vol)),
              "avg_pool2x2")
        levels.append(nxt)
        hl, wl = hl >> 1, wl >> 1
    return levels


def corr_pyramid_build_indexed(fmaps, idx1, idx2, levels=None, slots=None, layout=BLOCKED, num_levels=4, frame_range=None):
    """[fused] pyramids of the edges (idx1[e] -> idx2[e]) straight from the keyframe buffer: fmaps [n_frames,C,h,w] f16,
    idx1 / idx2 [E] int64 frame indices (factor_graph.py:147-148 gathers `fmaps[ii]`, `fmaps[jj]` first - never
    materialised here).  `levels`: existing level buffers of a pooled store, edge e is written to slot slots[e] (int32
    [E]); None: fresh buffers for E edges, slot e.  `frame_range` (lo, hi): the frames the indices lie in, for grids that
    need their operands prepared (callers with a host copy of the indices pass it; otherwise it is read back).  Returns
    the level list."""
    n, C, h, w = fmaps.shape
    E = int(idx1.shape[0])
    require(fused_build_covers(C, h, w, num_levels, fmaps.dtype) and fmaps.is_cuda,
            "corr_pyramid_build_indexed: fp16 maps on the device, C == 128, h, w >= 2^(levels-1)")
    check_gpu_contig(fmaps, idx1, idx2)
    require(idx1.dtype == torch.int64 and idx2.dtype == torch.int64 and idx2.shape[0] == E, "idx1 / idx2: int64 [E]")
    direct = blocked_dims(h, w)[5]
    require(direct or layout == BLOCKED, "the reference layout needs w % 64 == 0 and h % 8 == 0 (use corr_pyramid_build)")
    if levels is None:
        levels = [torch.empty(s, dtype=torch.float16, device=fmaps.device)
                  for s in pyramid_level_shapes(E, h, w, num_levels, layout)]
    else:
        want = pyramid_level_shapes(levels[0].shape[0], h, w, num_levels, layout)
        require(len(levels) == num_levels and all(tuple(lv.shape) == s and lv.is_contiguous() for lv, s in zip(levels, want)),
                "level buffers do not match the layout")
    if slots is not None:
        check_gpu_contig(slots)
        require(slots.dtype == torch.int32 and slots.shape[0] == E, "slots: int32 [E]")
    if E == 0:
        return levels
    sl = ptr(slots) if slots is not None else None
    if direct:
        check(lib().vipe_corr_pyramid_build_indexed(ptr(fmaps), ptr(idx1), ptr(idx2), sl, _level_ptrs(levels), E, C, h, w,
                                                    num_levels, layout, stream_ptr(fmaps)), "corr_pyramid_build_indexed")
        return levels
    if frame_range is None:
        both = torch.cat([idx1, idx2])
        frame_range = (int(both.min().item()), int(both.max().item()) + 1)
    lo, hi = max(0, int(frame_range[0])), min(n, int(frame_range[1]))
    require(hi > lo, "corr_pyramid_build_indexed: empty frame range")
    prep = corr_prep(fmaps[lo:hi])
    # an index outside [lo, hi) cannot be detected here without a read-back: the kernel skips such an edge (bounded read)
    check(lib().vipe_corr_pyramid_build_prepared(ptr(prep), lo, hi - lo, ptr(idx1), ptr(idx2), sl, _level_ptrs(levels), E, C, h, w,
                                                 num_levels, stream_ptr(fmaps)), "corr_pyramid_build_prepared")
    return levels


def corr_pyramid_lookup(levels, coords, radius=3):
    """CorrBlock.__call__ (droid_net.py:71-82) in ONE launch. levels: list of [E,h1,w1,h2>>i,w2>>i];
    coords [E,h1,w1,2] f32 -> [E, L*(2r+1)^2, h1, w1] in the volume dtype."""
    check_gpu_contig(coords, *levels)
    require(coords.dtype == torch.float32, "coords must be float32")
    E, h1, w1, h2, w2 = levels[0].shape
    L = len(levels)
    for i, lv in enumerate(levels):
        require(tuple(lv.shape) == (E, h1, w1, h2 >> i, w2 >> i) and lv.dtype == levels[0].dtype, "bad pyramid level")
    require(tuple(coords.shape) == (E, h1, w1, 2), "coords must be [E,h1,w1,2]")
    rd = 2 * radius + 1
    out = torch.empty((E, L * rd * rd, h1, w1), dtype=levels[0].dtype, device=coords.device)
    arr = (ctypes.c_void_p * L)(*[lv.data_ptr() for lv in levels])
    check(lib().vipe_corr_pyramid_lookup(ctypes.cast(arr, ctypes.c_void_p), ptr(coords), ptr(out), E, h1, w1, h2, w2,
                                         L, radius, DTYPE_CODE[levels[0].dtype], stream_ptr(coords)),
          "corr_pyramid_lookup")
    return out


def corr_pyramid_lookup_nhwc(levels, coords, radius=3, channel_stride=200):
    """Same lookup, written channels-last [E,h1,w1,channel_stride] (zero padded) for the MFMA convolutions."""
    check_gpu_contig(coords, *levels)
    require(coords.dtype == torch.float32, "coords must be float32")
    E, h1, w1, h2, w2 = levels[0].shape
    L = len(levels)
    require(tuple(coords.shape) == (E, h1, w1, 2), "coords must be [E,h1,w1,2]")
    out = torch.empty((E, h1, w1, channel_stride), dtype=levels[0].dtype, device=coords.device)
    arr = (ctypes.c_void_p * L)(*[lv.data_ptr() for lv in levels])
    check(lib().vipe_corr_pyramid_lookup_nhwc(ctypes.cast(arr, ctypes.c_void_p), ptr(coords), ptr(out), E, h1, w1, h2,
                                              w2, L, radius, DTYPE_CODE[levels[0].dtype], channel_stride,
                                              stream_ptr(coords)), "corr_pyramid_lookup_nhwc")
    return out


def corr_lookup_conv1x1(levels, coords, w_packed, bias, out, out_coff=0, cout=128, act="relu", slots=None, grid=None):
    """[fused] `CorrBlock.__call__` (4 levels, radius 3) + the correlation encoder's first 1x1 convolution
    (droid_net.py:436-437): writes out[E,h,w,C] channels [out_coff, out_coff + cout) without materialising the
    196-channel lookup.  levels: fp16 pyramid of `corr_pyramid_build` (reference layout) or of
    `corr_pyramid_build_indexed` (blocked layout: 7-D level 0; `grid` = (h, w) of the targets, needed when the store is
    padded); coords [E,h,w,2] f32.
    Raises NotImplementedError for configurations the fused kernel does not cover (callers fall back to
    `corr_pyramid_lookup_nhwc` + the conv)."""
    check_gpu_contig(coords, out, *levels)
    require(len(levels) == 4 and levels[0].dtype == torch.float16 and coords.dtype == torch.float32, "fp16 4-level pyramid")
    layout = BLOCKED if levels[0].dim() == 7 else REFERENCE
    cap = levels[2].shape[0]
    if levels[2].dim() == 5:
        h1, w1 = levels[2].shape[1:3]
        h2, w2 = levels[2].shape[3] << 2, levels[2].shape[4] << 2
    else:
        require(grid is not None, "a padded blocked store needs grid=(h, w)")
        h1, w1 = h2, w2 = int(grid[0]), int(grid[1])
    E = cap if slots is None else int(slots.shape[0])  # slots [E] int32: edge e reads pyramid slot slots[e] (pooled store)
    if slots is not None:
        check_gpu_contig(slots)
        require(slots.dtype == torch.int32 and E <= cap, "slots must be int32 [E], E <= pool capacity")
    require(tuple(coords.shape) == (E, h1, w1, 2) and tuple(out.shape[:3]) == (E, h1, w1) and out.dtype == torch.float16,
            "bad coords / out shape")
    arr = (ctypes.c_void_p * 4)(*[lv.data_ptr() for lv in levels])
    check(lib().vipe_corr_lookup_conv1x1(ctypes.cast(arr, ctypes.c_void_p), ptr(coords), ptr(w_packed), ptr(bias), ptr(out),
                                         out.shape[-1], out_coff, E, h1, w1, h2, w2, cout, {"none": 0, "relu": 1}[act],
                                         ptr(slots) if slots is not None else None, layout, stream_ptr(coords)),
          "corr_lookup_conv1x1")
    return out
