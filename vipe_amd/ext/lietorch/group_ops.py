"""Autograd-aware group operations (reference: vipe/ext/lietorch/group_ops.py): every op pairs the backend's forward
entry point with its backward entry point (`lietorch_ext.*_backward`, lietorch_gpu.cu:36-275).  Gradients of group
elements are tangent-space row vectors stored in the first K of the N embedding columns, exactly as in the reference,
so chains of these ops differentiate end to end; `FromVec` / `ToVec` convert to / from Euclidean gradients through
the orthogonal projector.  Used by `groups.LieGroup` whenever an input requires grad; under `torch.no_grad` (the
SLAM system, system.py:207) the ops call the backend directly."""

import torch

from .. import lietorch_ext as B


class GroupOp(torch.autograd.Function):
    forward_op = backward_op = None

    @classmethod
    def forward(cls, ctx, group_id, *inputs):
        ctx.group_id = group_id
        ctx.save_for_backward(*inputs)
        return cls.forward_op(group_id, *inputs)

    @classmethod
    def backward(cls, ctx, grad):
        assert cls.backward_op is not None, "Backward operation not implemented for {}".format(cls)
        out = cls.backward_op(ctx.group_id, grad.contiguous(), *ctx.saved_tensors)
        if isinstance(out, torch.Tensor):
            out = (out,)
        return (None,) + tuple(out)


def _op(name, fwd, bwd):
    return type(name, (GroupOp,), {"forward_op": staticmethod(fwd), "backward_op": None if bwd is None else staticmethod(bwd)})


Exp = _op("Exp", B.expm, B.expm_backward)
Log = _op("Log", B.logm, B.logm_backward)
Inv = _op("Inv", B.inv, B.inv_backward)
Mul = _op("Mul", B.mul, B.mul_backward)
Adj = _op("Adj", B.adj, B.adj_backward)
AdjT = _op("AdjT", B.adjT, B.adjT_backward)
Act3 = _op("Act3", B.act, B.act_backward)
Act4 = _op("Act4", B.act4, B.act4_backward)
Jinv = _op("Jinv", B.Jinv, None)
ToMatrix = _op("ToMatrix", B.as_matrix, None)


class FromVec(torch.autograd.Function):
    """Euclidean embedding -> group object (group_ops.py:95-108)."""

    @staticmethod
    def forward(ctx, group_id, x):
        ctx.group_id = group_id
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, grad):
        (x,) = ctx.saved_tensors
        J = B.projector(ctx.group_id, x.reshape(-1, x.shape[-1]).contiguous()).view(x.shape + (x.shape[-1],))
        return None, torch.matmul(grad.unsqueeze(-2), torch.linalg.pinv(J)).squeeze(-2)


class ToVec(torch.autograd.Function):
    """group object -> Euclidean embedding (group_ops.py:111-124)."""

    @staticmethod
    def forward(ctx, group_id, x):
        ctx.group_id = group_id
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, grad):
        (x,) = ctx.saved_tensors
        J = B.projector(ctx.group_id, x.reshape(-1, x.shape[-1]).contiguous()).view(x.shape + (x.shape[-1],))
        return None, torch.matmul(grad.unsqueeze(-2), J).squeeze(-2)
