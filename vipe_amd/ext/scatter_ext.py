"""scatter_ext: index-driven reductions along one dimension, autograd-aware (reference: csrc/scatter_ext/scatter.cpp:38-238
- five `torch::autograd::Function`s over `scatter_cuda`, cuda/scatter_cuda.cu:57-131).

Forward = ONE launch of the library's atomic scatter kernel (`vipe_scatter`, csrc/aux_ops.hip; `vipe_scatter_host` for
CPU tensors) on `src` viewed as [outer, src_dim, inner]; a second pass fills the arg indices of min / max.  All five
reductions share one Function whose backward follows the table below (the adjoints of scatter.cpp:54-62, 85-95,
134-145, 173-185):

    sum    dL/dsrc = gather(g, index)
    mean   dL/dsrc = gather(g / count, index)            count = max(#rows scattered to the slot, 1)
    mul    dL/dsrc = gather(g * out, index) / src        (0 where that is nan)
    min/max dL/dsrc[e] = g[slot] where e is the slot's arg row, else 0
"""

import torch

from .._lib import DTYPE_CODE, check, lib, ptr, require, stream_ptr

_REDUCE = {"sum": 0, "mul": 1, "mean": 2, "min": 3, "max": 4}
_FILL = {"sum": 0.0, "mean": 0.0, "mul": 1.0, "min": float("inf"), "max": float("-inf")}


def _expand_index(index, src, dim):
    """index -> src's shape (scatter.cpp:19-26): a 1-D index runs along `dim`, missing trailing dims are broadcast."""
    if index.dim() == 1 and src.dim() > 1:
        index = index.view((1,) * dim + (-1,))
    index = index.view(tuple(index.shape) + (1,) * (src.dim() - index.dim()))
    return index.expand(src.shape).contiguous()


def _launch(src, index, out, arg, dim, reduce):
    outer = 1
    for s in src.shape[:dim]:
        outer *= int(s)
    inner = 1
    for s in src.shape[dim + 1:]:
        inner *= int(s)
    L = lib()
    if src.is_cuda:
        check(L.vipe_scatter(ptr(src), ptr(index), ptr(out), ptr(arg), outer, src.shape[dim], inner, out.shape[dim],
                             _REDUCE[reduce], DTYPE_CODE[src.dtype], stream_ptr(src)), "scatter_" + reduce)
    else:
        require(src.dtype in (torch.float32, torch.float64), "scatter on CPU tensors: float32 / float64")
        check(L.vipe_scatter_host(ptr(src), ptr(index), ptr(out), ptr(arg), outer, src.shape[dim], inner, out.shape[dim],
                                  _REDUCE[reduce], DTYPE_CODE[src.dtype]), "scatter_" + reduce)


class _Scatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, dim, out, dim_size, reduce):
        require(src.dtype in DTYPE_CODE, "scatter: half / float / double")
        require(index.dtype == torch.int64 and index.device == src.device, "index: int64 on src's device")
        dim = dim + src.dim() if dim < 0 else dim
        src_c = src.contiguous()
        index_x = _expand_index(index, src_c, dim)
        fresh = out is None
        if fresh:
            sizes = list(src.shape)
            if dim_size is not None:
                sizes[dim] = int(dim_size)
            else:  # scatter_cuda.cu:80-85 - the one host read-back of this op
                sizes[dim] = int(index_x.max()) + 1 if index_x.numel() else 0
            out = torch.full(sizes, _FILL[reduce], dtype=src.dtype, device=src.device)
        else:
            require(out.is_contiguous() and out.dtype == src.dtype and out.device == src.device, "out: contiguous, like src")
            ctx.mark_dirty(out)
        arg = None
        if reduce in ("min", "max"):
            arg = torch.full(out.shape, src.shape[dim], dtype=torch.int64, device=src.device)
        _launch(src_c, index_x, out, arg, dim, reduce)
        count = None
        if reduce == "mean":  # rows per slot through the same kernel (scatter.cpp:117-122)
            count = torch.zeros(out.shape, dtype=src.dtype, device=src.device)
            _launch(torch.ones_like(src_c), index_x, count, None, dim, "sum")
            count.clamp_(min=1)
            out.div_(count)
        if arg is not None:
            if fresh:
                out.masked_fill_(arg == src.shape[dim], 0)  # scatter_cuda.cu:141-142: untouched slots read 0
            ctx.mark_non_differentiable(arg)
        ctx.dim, ctx.reduce, ctx.n_src = dim, reduce, src.shape[dim]
        ctx.save_for_backward(index_x, *(t for t in {"sum": (), "mean": (count,), "mul": (src_c, out),
                                                    "min": (arg,), "max": (arg,)}[reduce]))
        return (out, arg) if arg is not None else out

    @staticmethod
    def backward(ctx, g, *_):
        index, *saved = ctx.saved_tensors
        dim, reduce = ctx.dim, ctx.reduce
        if reduce == "sum":
            gi = g.gather(dim, index)
        elif reduce == "mean":
            gi = (g / saved[0]).gather(dim, index)
        elif reduce == "mul":
            src, out = saved
            gi = (g * out).gather(dim, index) / src
            gi = gi.masked_fill(gi.isnan(), 0)
        else:  # the slot's gradient goes to its arg row; slots nothing reached point one past the end
            shape = list(index.shape)
            shape[dim] = ctx.n_src + 1
            gi = torch.zeros(shape, dtype=g.dtype, device=g.device).scatter_(dim, saved[0], g).narrow(dim, 0, ctx.n_src)
        return gi, None, None, None, None, None


def scatter_sum(src, index, dim, out=None, dim_size=None):
    return _Scatter.apply(src, index, dim, out, dim_size, "sum")


def scatter_mul(src, index, dim, out=None, dim_size=None):
    return _Scatter.apply(src, index, dim, out, dim_size, "mul")


def scatter_mean(src, index, dim, out=None, dim_size=None):
    return _Scatter.apply(src, index, dim, out, dim_size, "mean")


def scatter_min(src, index, dim, out=None, dim_size=None):
    return _Scatter.apply(src, index, dim, out, dim_size, "min")


def scatter_max(src, index, dim, out=None, dim_size=None):
    return _Scatter.apply(src, index, dim, out, dim_size, "max")
