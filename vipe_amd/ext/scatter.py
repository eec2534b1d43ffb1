"""Python-level scatter API (what `vipe.slam` / `vipe.priors` host code imports from `vipe.ext.scatter`:
scatter_add / scatter_sum / scatter_mean / scatter_mul / scatter_min / scatter_max / scatter, vipe/ext/scatter.py:24-110).

Every reduction - sum and mean included, which the reference sends through `Tensor.scatter_add_` - runs in the library's
own scatter kernel via `scatter_ext` (autograd-aware): GraphAgg's scatter_mean (droid_net.py:420-421) is a factor-graph
scatter-add, one of the operators this backend exists to provide."""

from . import scatter_ext

_OPS = {"sum": scatter_ext.scatter_sum, "add": scatter_ext.scatter_sum, "mul": scatter_ext.scatter_mul,
        "mean": scatter_ext.scatter_mean}


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    return scatter_ext.scatter_sum(src, index, dim, out, dim_size)


scatter_add = scatter_sum


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    return scatter_ext.scatter_mean(src, index, dim, out, dim_size)


def scatter_mul(src, index, dim=-1, out=None, dim_size=None):
    return scatter_ext.scatter_mul(src, index, dim, out, dim_size)


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    return scatter_ext.scatter_min(src, index, dim, out, dim_size)


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    return scatter_ext.scatter_max(src, index, dim, out, dim_size)


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    """One entry point for all reductions; min / max return the values only (vipe/ext/scatter.py:200-203)."""
    if reduce in _OPS:
        return _OPS[reduce](src, index, dim, out, dim_size)
    if reduce in ("min", "max"):
        return getattr(scatter_ext, "scatter_" + reduce)(src, index, dim, out, dim_size)[0]
    raise ValueError(f"unknown reduction {reduce!r}")
