"""LieGroup Python API (reference: vipe/ext/lietorch/groups.py:54-328) over `lietorch_ext`.

Under torch.no_grad (the SLAM system, system.py:207) ops call the backend directly; when an input requires grad they
go through the autograd Functions of `group_ops` (forward + backward entry points of the C ABI).  Broadcasting follows broadcasting.py:14-41, except that a group element shared by
whole trailing blocks of rows is NOT replicated per row on the device (adjT / act4 use the *_bcast
entry points of the C ABI).
"""

import numpy as np
import torch

from .. import lietorch_ext as B
from . import group_ops as GO

_AUTOGRAD = {B.expm: GO.Exp, B.logm: GO.Log, B.inv: GO.Inv, B.mul: GO.Mul, B.adj: GO.Adj, B.adjT: GO.AdjT,
             B.act: GO.Act3, B.act4: GO.Act4, B.Jinv: GO.Jinv}


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t.requires_grad for t in ts)


def _broadcast(x, y):
    """-> (x2d, y2d, out_shape, rows_per_elem or None)."""
    assert x.dim() == y.dim(), "lietorch broadcasting needs equal rank (broadcasting.py:9-12)"
    xs, ys = x.shape[:-1], y.shape[:-1]
    for n, m in zip(xs, ys):
        assert n == m or n == 1 or m == 1
    out_shape = tuple(max(n, m) for n, m in zip(xs, ys))
    if xs == ys:
        return x.reshape(-1, x.shape[-1]).contiguous(), y.reshape(-1, y.shape[-1]).contiguous(), out_shape, None
    # fast path: x = [lead..., 1, 1, ...], y = [lead..., trailing...]  -> one element per block of rows
    k = len(xs)
    while k > 0 and xs[k - 1] == 1:
        k -= 1
    if xs[:k] == ys[:k] and y.is_cuda:
        rows = int(np.prod(ys[k:])) if k < len(ys) else 1
        return x.reshape(-1, x.shape[-1]).contiguous(), y.reshape(-1, y.shape[-1]).contiguous(), out_shape, rows
    xe = x.expand(out_shape + (x.shape[-1],)).reshape(-1, x.shape[-1]).contiguous()
    ye = y.expand(out_shape + (y.shape[-1],)).reshape(-1, y.shape[-1]).contiguous()
    return xe, ye, out_shape, None


class LieGroup:
    group_name = group_id = manifold_dim = embedded_dim = id_elem = None

    def __init__(self, data):
        self.data = data

    def __repr__(self):
        return "{}: size={}, device={}, dtype={}".format(self.group_name, self.shape, self.device, self.dtype)

    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def tangent_shape(self):
        return self.data.shape[:-1] + (self.manifold_dim,)

    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        if isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        data = cls.id_elem.reshape(1, -1)
        if "device" in kwargs:
            data = data.to(kwargs["device"])
        if "dtype" in kwargs:
            data = data.type(kwargs["dtype"])
        data = data.repeat(int(np.prod(batch_shape)), 1)
        return cls(data).view(tuple(batch_shape))

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(G.shape, device=G.data.device, dtype=G.data.dtype)

    @classmethod
    def Random(cls, *batch_shape, sigma=1.0, **kwargs):
        if isinstance(batch_shape[0], (tuple, list)):
            batch_shape = tuple(batch_shape[0])
        return cls.exp(sigma * torch.randn(tuple(batch_shape) + (cls.manifold_dim,), **kwargs))

    @classmethod
    def _un(cls, fn, x):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        if _needs_grad(x):
            return _AUTOGRAD[fn].apply(cls.group_id, x2).view(x.shape[:-1] + (-1,))
        return fn(cls.group_id, x2).view(x.shape[:-1] + (-1,))

    @classmethod
    def _bin(cls, fn, x, y, bcast_fn=None):
        if _needs_grad(x, y):  # differentiable path: plain expansion (the reference's broadcasting.py), autograd ops
            out_shape = tuple(max(n, m) for n, m in zip(x.shape[:-1], y.shape[:-1]))
            xe = x.expand(out_shape + (x.shape[-1],)).reshape(-1, x.shape[-1]).contiguous()
            ye = y.expand(out_shape + (y.shape[-1],)).reshape(-1, y.shape[-1]).contiguous()
            return _AUTOGRAD[fn].apply(cls.group_id, xe, ye).view(out_shape + (-1,))
        x2, y2, out_shape, rows = _broadcast(x, y)
        if rows is not None:
            if bcast_fn is not None:
                return bcast_fn(cls.group_id, x2, y2).view(out_shape + (-1,))
            x2 = x2.repeat_interleave(rows, dim=0)
        return fn(cls.group_id, x2, y2).view(out_shape + (-1,))

    @classmethod
    def InitFromVec(cls, data):
        """groups.py:104-106: Euclidean embedding -> group (gradient through the projector's pseudo-inverse)."""
        return cls(GO.FromVec.apply(cls.group_id, data) if _needs_grad(data) else data)

    def vec(self):
        """groups.py:201-202: group -> Euclidean embedding."""
        return GO.ToVec.apply(self.group_id, self.data) if _needs_grad(self.data) else self.data

    @classmethod
    def exp(cls, x):
        return cls(cls._un(B.expm, x))

    def log(self):
        return self._un(B.logm, self.data)

    def inv(self):
        return self.__class__(self._un(B.inv, self.data))

    def mul(self, other):
        return self.__class__(self._bin(B.mul, self.data, other.data))

    def retr(self, a):
        """retraction: Exp(a) * X (groups.py:147-150)"""
        return self.__class__(self._bin(B.mul, self._un(B.expm, a), self.data))

    def adj(self, a):
        return self._bin(B.adj, self.data, a)

    def adjT(self, a):
        return self._bin(B.adjT, self.data, a, B.adjT_bcast)

    def Jinv(self, a):
        return self._bin(B.Jinv, self.data, a)

    def act(self, p):
        if p.shape[-1] == 3:
            return self._bin(B.act, self.data, p)
        return self._bin(B.act4, self.data, p, B.act4_bcast)

    def matrix(self):
        return B.as_matrix(self.group_id, self.data.reshape(-1, self.embedded_dim).contiguous()).view(self.shape + (4, 4))

    def translation(self):
        p = torch.as_tensor([0.0, 0.0, 0.0, 1.0], dtype=self.dtype, device=self.device)
        return self.act(p.view([1] * (self.data.dim() - 1) + [4]).expand(self.shape + (4,)))

    def quaternion(self):
        off = 0 if self.group_id in (1, 2) else 3
        return self.data[..., off:off + 4]

    def detach(self):
        return self.__class__(self.data.detach())

    def view(self, dims):
        return self.__class__(self.data.view(tuple(dims) + (self.embedded_dim,)))

    def __mul__(self, other):
        if isinstance(other, LieGroup):
            return self.mul(other)
        if isinstance(other, torch.Tensor):
            return self.act(other)
        return NotImplemented

    def __getitem__(self, index):
        return self.__class__(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def to(self, *args, **kwargs):
        return self.__class__(self.data.to(*args, **kwargs))

    def cpu(self):
        return self.__class__(self.data.cpu())

    def cuda(self):
        return self.__class__(self.data.cuda())

    def float(self, device=None):
        return self.__class__(self.data.float())

    def double(self, device=None):
        return self.__class__(self.data.double())

    def unbind(self, dim=0):
        return [self.__class__(x) for x in self.data.unbind(dim=dim)]


class SO3(LieGroup):
    group_name, group_id, manifold_dim, embedded_dim = "SO3", 1, 3, 4
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 1.0])

    def __init__(self, data):
        if isinstance(data, SE3):
            data = data.data[..., 3:7]
        super().__init__(data)


class RxSO3(LieGroup):
    group_name, group_id, manifold_dim, embedded_dim = "RxSO3", 2, 4, 5
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 1.0, 1.0])

    def __init__(self, data):
        if isinstance(data, Sim3):
            data = data.data[..., 3:8]
        super().__init__(data)


class SE3(LieGroup):
    group_name, group_id, manifold_dim, embedded_dim = "SE3", 3, 6, 7
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])

    def __init__(self, data):
        if isinstance(data, SO3):
            data = torch.cat([torch.zeros_like(data.data[..., :3]), data.data], -1)
        super().__init__(data)

    def scale(self, s):
        t, q = self.data.split([3, 4], -1)
        return SE3(torch.cat([t * s.unsqueeze(-1), q], dim=-1))


class Sim3(LieGroup):
    group_name, group_id, manifold_dim, embedded_dim = "Sim3", 4, 7, 8
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 1.0])

    def __init__(self, data):
        if isinstance(data, SE3):
            data = torch.cat([data.data, torch.ones_like(data.data[..., :1])], -1)
        elif isinstance(data, Sim3):
            data = data.data
        super().__init__(data)


def cat(group_list, dim):
    return group_list[0].__class__(torch.cat([X.data for X in group_list], dim=dim))


def stack(group_list, dim):
    return group_list[0].__class__(torch.stack([X.data for X in group_list], dim=dim))
