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

import os

import torch

from .._lib import DTYPE_CODE, check, lib, ptr, require, stream_ptr

_REDUCE = {"sum": 0, "mul": 1, "mean": 2, "min": 3, "max": 4}
_FILL = {"sum": 0.0, "mean": 0.0, "mul": 1.0, "min": float("inf"), "max": float("-inf")}


def _expand_index(index, src, dim):
    """index -> src's shape (scatter.cpp:19-26): a 1-D index runs along `dim`, missing trailing dims are broadcast."""
    if index.dim() == 1 and src.dim() > 1:
        index = index.view((1,) * dim + (-1,))
    index = index.view(tuple(index.shape) + (1,) * (src.dim() - index.dim()))
    return index.expand(src.shape)


def _row_index(index, src, dim):
    """the index as one slot per row of `dim` ([src.shape[dim]] int64) when that is all it says - a 1-D index, or one
    whose other dimensions are 1 (what scatter_mean / scatter_add callers pass: droid_net.py:420-421) - else None"""
    if index.dim() == 1:
        return index.contiguous() if index.shape[0] == src.shape[dim] else None
    if dim >= index.dim():  # scatter.cpp:19-26 appends trailing dims: the index does not run along `dim` at all
        return None
    if index.dim() <= src.dim() and all(int(s) == 1 for i, s in enumerate(index.shape) if i != dim) and index.shape[dim] == src.shape[dim]:
        return index.reshape(-1).contiguous()
    return None


def _launch(src, index, out, arg, dim, reduce, rows):
    outer = 1
    for s in src.shape[:dim]:
        outer *= int(s)
    inner = 1
    for s in src.shape[dim + 1:]:
        inner *= int(s)
    L = lib()
    if src.is_cuda:
        if os.environ.get("VIPE_AMD_CHECK_INDICES") and index.numel():
            # The kernels SKIP slots outside [0, out.shape[dim]) instead of writing out of bounds; torch's scatter and the
            # reference's scatter_cuda assert.  This opt-in check (a device-to-host sync) turns a wrong dim_size or a
            # corrupted index into the error the host path raises, so that forward and backward cannot quietly disagree.
            lo, hi = int(index.min()), int(index.max())
            require(0 <= lo and hi < out.shape[dim], f"scatter index out of range: [{lo}, {hi}] vs dim size {out.shape[dim]}")
        fn = L.vipe_scatter_rows if rows else L.vipe_scatter
        check(fn(ptr(src), ptr(index), ptr(out), ptr(arg), outer, src.shape[dim], inner, out.shape[dim],
                 _REDUCE[reduce], DTYPE_CODE[src.dtype], stream_ptr(src)), "scatter_" + reduce)
    else:
        require(src.dtype in (torch.float32, torch.float64), "scatter on CPU tensors: float32 / float64")
        if rows:
            index = _expand_index(index, src, dim).contiguous()
        check(L.vipe_scatter_host(ptr(src), ptr(index), ptr(out), ptr(arg), outer, src.shape[dim], inner, out.shape[dim],
                                  _REDUCE[reduce], DTYPE_CODE[src.dtype]), "scatter_" + reduce)


class _Scatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, dim, out, dim_size, reduce):
        require(src.dtype in DTYPE_CODE, "scatter: half / float / double")
        require(index.dtype == torch.int64 and index.device == src.device, "index: int64 on src's device")
        dim = dim + src.dim() if dim < 0 else dim
        src_c = src.contiguous()
        # one slot per row of `dim` (the usual call): the kernel takes the [E] index as it is; a general index is expanded
        # to src's shape as the reference does (scatter.cpp:19-26)
        idx = _row_index(index, src_c, dim)
        rows = idx is not None
        if not rows:
            idx = _expand_index(index, src_c, dim).contiguous()
        fresh = out is None
        if fresh:
            sizes = list(src.shape)
            if dim_size is not None:
                sizes[dim] = int(dim_size)
            else:  # scatter_cuda.cu:80-85 - the one host read-back of this op
                sizes[dim] = int(idx.max()) + 1 if idx.numel() else 0
            out = torch.full(sizes, _FILL[reduce], dtype=src.dtype, device=src.device)
        else:
            require(out.is_contiguous() and out.dtype == src.dtype and out.device == src.device, "out: contiguous, like src")
            ctx.mark_dirty(out)
        arg = None
        if reduce in ("min", "max"):
            arg = torch.full(out.shape, src.shape[dim], dtype=torch.int64, device=src.device)
        _launch(src_c, idx, out, arg, dim, reduce, rows)
        count = None
        if reduce == "mean":  # rows per slot through the same kernel (scatter.cpp:117-122)
            if rows:  # counted on the index's own shape: a vector [out_dim], broadcast along every other dimension
                cnt = torch.zeros(out.shape[dim], dtype=torch.float32, device=src.device)
                _launch(torch.ones(src.shape[dim], dtype=torch.float32, device=src.device), idx, cnt, None, 0, "sum", True)
                count = cnt.clamp_(min=1).to(src.dtype).view((1,) * dim + (-1,) + (1,) * (src.dim() - dim - 1))
            else:
                count = torch.zeros(out.shape, dtype=src.dtype, device=src.device)
                _launch(torch.ones_like(src_c), idx, count, None, dim, "sum", False)
                count.clamp_(min=1)
            out.div_(count)
        if arg is not None:
            if fresh:
                out.masked_fill_(arg == src.shape[dim], 0)  # scatter_cuda.cu:141-142: untouched slots read 0
            ctx.mark_non_differentiable(arg)
        ctx.dim, ctx.reduce, ctx.n_src, ctx.rows, ctx.src_shape = dim, reduce, src.shape[dim], rows, tuple(src.shape)
        ctx.save_for_backward(idx, *(t for t in {"sum": (), "mean": (count,), "mul": (src_c, out),
                                                 "min": (arg,), "max": (arg,)}[reduce]))
        return (out, arg) if arg is not None else out

    @staticmethod
    def backward(ctx, g, *_):
        index, *saved = ctx.saved_tensors
        dim, reduce = ctx.dim, ctx.reduce
        if ctx.rows:  # a stride-0 view of the [E] index in src's shape: gather reads it without a copy
            index = index.view((1,) * dim + (-1,) + (1,) * (len(ctx.src_shape) - dim - 1)).expand(ctx.src_shape)
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


def _integer_scatter(src, index, dim, out, dim_size, mean):
    """integer sources (the reference's Python layer accepts them through Tensor.scatter_add_, vipe/ext/scatter.py:24-63;
    the atomic kernel covers half / float / double): torch's own scatter, no gradient to carry"""
    dim = dim + src.dim() if dim < 0 else dim
    idx = _expand_index(index, src, dim)
    if out is None:
        sizes = list(src.shape)
        sizes[dim] = int(dim_size) if dim_size is not None else (int(idx.max()) + 1 if idx.numel() else 0)
        out = torch.zeros(sizes, dtype=src.dtype, device=src.device)
    out.scatter_add_(dim, idx, src)
    if mean:
        cnt = torch.zeros_like(out).scatter_add_(dim, idx, torch.ones_like(src)).clamp_(min=1)
        out.copy_(torch.div(out, cnt, rounding_mode="floor"))
    return out


def scatter_sum(src, index, dim, out=None, dim_size=None):
    if not src.is_floating_point():
        return _integer_scatter(src, index, dim, out, dim_size, False)
    return _Scatter.apply(src, index, dim, out, dim_size, "sum")


def scatter_mul(src, index, dim, out=None, dim_size=None):
    return _Scatter.apply(src, index, dim, out, dim_size, "mul")


def scatter_mean(src, index, dim, out=None, dim_size=None):
    if not src.is_floating_point():
        return _integer_scatter(src, index, dim, out, dim_size, True)
    return _Scatter.apply(src, index, dim, out, dim_size, "mean")


def scatter_min(src, index, dim, out=None, dim_size=None):
    return _Scatter.apply(src, index, dim, out, dim_size, "min")


def scatter_max(src, index, dim, out=None, dim_size=None):
    return _Scatter.apply(src, index, dim, out, dim_size, "max")
