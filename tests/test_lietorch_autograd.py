"""Backward passes of lietorch_ext (lietorch_gpu.cu:36-275, so3.h / se3.h Jacobians) for SO3 and SE3.

The reference's own check for these is lietorch's gradcheck of op chains with Euclidean inputs and outputs: group
gradients are tangent-space row vectors, so only whole chains are comparable with finite differences.  The forward
ops are pinned against the closed-form oracle elsewhere; here torch.autograd.gradcheck (float64) pins every backward
entry point, the projector (through `vec()`) and Jinv (against the dense Jacobian).  CPU tensors run the library's
host loop (the reference ships lietorch_cpu.cpp); the GPU variant runs the same functors in the HIP kernel."""
import pytest
import torch

from vipe_amd.ext.lietorch import SE3, SO3


def _chains(G):
    K = G.manifold_dim
    return {
        "exp_log": (lambda a: G.exp(a).log(), [K]),
        "mul": (lambda a, b: (G.exp(a) * G.exp(b)).log(), [K, K]),
        "inv": (lambda a: G.exp(a).inv().log(), [K]),
        "adj": (lambda a, v: G.exp(a).adj(v), [K, K]),
        "adjT": (lambda a, v: G.exp(a).adjT(v), [K, K]),
        "act3": (lambda a, p: G.exp(a).act(p), [K, 3]),
        "act4": (lambda a, p: G.exp(a).act(p), [K, 4]),
        "retr_chain": (lambda a, b, p: (G.exp(a).retr(b).inv() * G.exp(b)).act(p), [K, K, 3]),
        "vec": (lambda a: G.exp(a).vec(), [K]),
    }


def _run(G, name, device):
    fn, dims = _chains(G)[name]
    gen = torch.Generator().manual_seed(hash(name) % 1000)
    inputs = [(0.7 * torch.randn(5, d, generator=gen, dtype=torch.float64)).to(device).requires_grad_(True) for d in dims]
    assert torch.autograd.gradcheck(fn, inputs, eps=1e-6, atol=1e-7, rtol=1e-6)


@pytest.mark.parametrize("G", [SO3, SE3], ids=["SO3", "SE3"])
@pytest.mark.parametrize("name", list(_chains(SE3)))
def test_backward_ops_gradcheck_host(G, name):
    _run(G, name, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("G", [SO3, SE3], ids=["SO3", "SE3"])
@pytest.mark.parametrize("name", list(_chains(SE3)))
def test_backward_ops_gradcheck_device(G, name):
    _run(G, name, torch.device("cuda:0"))


@pytest.mark.parametrize("G", [SO3, SE3], ids=["SO3", "SE3"])
def test_jinv_is_inverse_left_jacobian_of_log(G):
    """Jinv(X, a) = Jl^-1(log X) a (lietorch_gpu.cu:263-275): Jl^-1 is d log(exp(e) X) / d e at e = 0."""
    gen = torch.Generator().manual_seed(3)
    K = G.manifold_dim
    x = 0.6 * torch.randn(4, K, generator=gen, dtype=torch.float64)
    X = G.exp(x)
    a = torch.randn(4, K, generator=gen, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(lambda e: G.exp(e).mul(G(X.data)).log(), torch.zeros(4, K, dtype=torch.float64))
    Jd = torch.stack([J[i, :, i, :] for i in range(4)])  # [4, K, K]
    assert torch.allclose(X.Jinv(a), torch.einsum("nij,nj->ni", Jd, a), atol=1e-8)
