"""lietorch_ext: batched Lie-group ops (reference: csrc/lietorch_ext/lietorch.cpp:305-336).

Same names and (group_id, tensors...) calling convention.  CPU tensors run the library's host loop over
the same closed forms (the reference ships lietorch_cpu.cpp); CUDA/HIP tensors launch on the current stream.
float32 / float64 only - other dtypes raise (the reference silently returns garbage, dispatch.h:42-52).
"""

import torch

from .._lib import DTYPE_CODE, check, lib, ptr, require, stream_ptr

_N = {1: 4, 2: 5, 3: 7, 4: 8}
_K = {1: 3, 2: 4, 3: 6, 4: 7}


def _prep(*ts):
    dev = ts[0].device
    for t in ts:
        require(t.device == dev, "all inputs must be on the same device")
        require(t.dtype in (torch.float32, torch.float64) and t.dtype == ts[0].dtype, "float32/float64 inputs of one dtype")
        require(t.is_contiguous() and t.dim() == 2, "inputs must be contiguous [n, dim]")
    return DTYPE_CODE[ts[0].dtype], int(dev.type == "cuda"), (stream_ptr(ts[0]) if dev.type == "cuda" else None)


def _unary(fn, gid, x, out_dim):
    dt, on_dev, st = _prep(x)
    out = torch.empty((x.shape[0], out_dim), dtype=x.dtype, device=x.device)
    check(fn(gid, ptr(x), ptr(out), x.shape[0], dt, on_dev, st), fn.__name__)
    return out


def _binary(fn, gid, x, y, out_dim):
    dt, on_dev, st = _prep(x, y)
    require(x.shape[0] == y.shape[0], "row counts differ (broadcast on the Python side first)")
    out = torch.empty((x.shape[0], out_dim), dtype=x.dtype, device=x.device)
    check(fn(gid, ptr(x), ptr(y), ptr(out), x.shape[0], dt, on_dev, st), fn.__name__)
    return out


def expm(gid, a):
    return _unary(lib().vipe_lie_expm, gid, a, _N[gid])


def logm(gid, X):
    return _unary(lib().vipe_lie_logm, gid, X, _K[gid])


def inv(gid, X):
    return _unary(lib().vipe_lie_inv, gid, X, _N[gid])


def mul(gid, X, Y):
    return _binary(lib().vipe_lie_mul, gid, X, Y, _N[gid])


def adj(gid, X, a):
    return _binary(lib().vipe_lie_adj, gid, X, a, _K[gid])


def adjT(gid, X, a):
    return _binary(lib().vipe_lie_adjT, gid, X, a, _K[gid])


def act(gid, X, p):
    return _binary(lib().vipe_lie_act, gid, X, p, 3)


def act4(gid, X, p):
    return _binary(lib().vipe_lie_act4, gid, X, p, 4)


def as_matrix(gid, X):
    return _unary(lib().vipe_lie_as_matrix, gid, X, 16).view(-1, 4, 4)


def projector(gid, X):
    return _unary(lib().vipe_lie_projector, gid, X, _N[gid] * _N[gid]).view(-1, _N[gid], _N[gid])


def Jinv(gid, X, a):
    return _binary(lib().vipe_lie_jinv, gid, X, a, _K[gid])


def adjT_bcast(gid, X, a):
    """X [n,N] shared by the r = a.shape[0]/n consecutive rows of a (no replication of X, cf. broadcasting.py:32-35)."""
    dt, on_dev, st = _prep(X, a)
    require(on_dev == 1 and a.shape[0] % X.shape[0] == 0, "adjT_bcast: device tensors, rows divisible")
    out = torch.empty_like(a)
    check(lib().vipe_lie_adjT_bcast(gid, ptr(X), ptr(a), ptr(out), X.shape[0], a.shape[0] // X.shape[0], dt, st), "adjT_bcast")
    return out


def act4_bcast(gid, X, p):
    dt, on_dev, st = _prep(X, p)
    require(on_dev == 1 and p.shape[0] % X.shape[0] == 0, "act4_bcast: device tensors, rows divisible")
    out = torch.empty_like(p)
    check(lib().vipe_lie_act4_bcast(gid, ptr(X), ptr(p), ptr(out), X.shape[0], p.shape[0] // X.shape[0], dt, st), "act4_bcast")
    return out


def _bwd(name):
    def f(gid, grad, *inputs):
        fn = getattr(lib(), "vipe_lie_" + name + "_backward")
        dt, on_dev, st = _prep(grad, *inputs)
        outs = [torch.zeros_like(t) for t in inputs]
        check(fn(gid, ptr(grad), *[ptr(t) for t in inputs], *[ptr(o) for o in outs], grad.shape[0], dt, on_dev, st),
              name + "_backward")
        return outs if len(outs) > 1 else outs[0]
    return f


expm_backward = _bwd("expm")
logm_backward = _bwd("logm")
inv_backward = _bwd("inv")
mul_backward = _bwd("mul")
adj_backward = _bwd("adj")
adjT_backward = _bwd("adjT")
act_backward = _bwd("act")
act4_backward = _bwd("act4")
