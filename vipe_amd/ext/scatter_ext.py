"""scatter_ext: atomic scatter reductions (reference: csrc/scatter_ext/scatter.cpp:232-238)."""

import torch

from .._lib import DTYPE_CODE, check, check_gpu_contig, lib, ptr, require, stream_ptr

_REDUCE = {"sum": 0, "mul": 1, "mean": 2, "min": 3, "max": 4}


def _broadcast(index, src, dim):
    """scatter.cpp:13-26."""
    if dim < 0:
        dim += src.dim()
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size()).contiguous(), dim


def _scatter(src, index, dim, out, dim_size, reduce):
    check_gpu_contig(src)
    require(src.dtype in DTYPE_CODE, "scatter: half/float/double")
    index, dim = _broadcast(index, src.contiguous(), dim)
    sizes = list(src.shape)
    if out is None:
        if dim_size is not None:
            sizes[dim] = int(dim_size)
        elif index.numel() == 0:
            sizes[dim] = 0
        else:
            sizes[dim] = int(index.max()) + 1
        fill = {"sum": 0, "mean": 0, "mul": 1, "min": float("inf"), "max": float("-inf")}[reduce]
        out = torch.full(sizes, fill, dtype=src.dtype, device=src.device)
        fresh = True
    else:
        require(out.is_contiguous(), "out must be contiguous")
        fresh = False
    outer = 1
    for s in src.shape[:dim]:
        outer *= s
    inner = 1
    for s in src.shape[dim + 1:]:
        inner *= s
    arg = None
    if reduce in ("min", "max"):
        arg = torch.full(out.shape, src.shape[dim], dtype=torch.int64, device=src.device)
    check(lib().vipe_scatter(ptr(src), ptr(index), ptr(out), ptr(arg), outer, src.shape[dim], inner, out.shape[dim],
                             _REDUCE[reduce], DTYPE_CODE[src.dtype], stream_ptr(src)), "scatter_" + reduce)
    if reduce in ("min", "max") and fresh:
        out.masked_fill_(arg == src.shape[dim], 0)  # scatter.cpp:141-142: untouched entries become 0
    return out, arg, index, dim


def scatter_sum(src, index, dim, out=None, dim_size=None):
    return _scatter(src, index, dim, out, dim_size, "sum")[0]


def scatter_mul(src, index, dim, out=None, dim_size=None):
    return _scatter(src, index, dim, out, dim_size, "mul")[0]


def scatter_mean(src, index, dim, out=None, dim_size=None):
    o, _, idx, d = _scatter(src, index, dim, out, dim_size, "sum")
    cnt = torch.zeros(o.shape, dtype=src.dtype, device=src.device)
    ones = torch.ones_like(src)
    check(lib().vipe_scatter(ptr(ones), ptr(idx), ptr(cnt), None, int(torch.tensor(src.shape[:d]).prod()) if d else 1,
                             src.shape[d], int(torch.tensor(src.shape[d + 1:]).prod()) if d + 1 < src.dim() else 1,
                             o.shape[d], 0, DTYPE_CODE[src.dtype], stream_ptr(src)), "scatter_mean(count)")
    cnt.clamp_(min=1)
    return o.div_(cnt) if o.is_floating_point() else o.div_(cnt, rounding_mode="floor")


def scatter_min(src, index, dim, out=None, dim_size=None):
    o, a, _, _ = _scatter(src, index, dim, out, dim_size, "min")
    return o, a


def scatter_max(src, index, dim, out=None, dim_size=None):
    o, a, _, _ = _scatter(src, index, dim, out, dim_size, "max")
    return o, a
