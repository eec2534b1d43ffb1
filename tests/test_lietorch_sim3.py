"""RxSO3 / Sim3 entries of lietorch_ext (group ids 2 and 4; csrc/lietorch_ext/rxso3.h, sim3.h).

PARITY UNPINNED by the reference (no test, no fixture, its lietorch needs Eigen + CUDA): the forward ops are pinned by
float64 group identities against an independent formulation - the 4x4 matrix exponential of hat(a)
(scipy.linalg.expm) - and the backward passes by torch.autograd.gradcheck of op chains.  Sim3's left Jacobian and its
inverse are TRUNCATED series in the reference (sim3.h:165-184), restated as written, so the gradient checks for Sim3
use small tangents where the truncation error (|a|^5 / 720) is below the check's tolerance."""
import numpy as np
import pytest
import scipy.linalg
import torch

from vipe_amd.ext.lietorch import SE3, RxSO3, Sim3


def _hat3(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], np.float64)


def _hat(G, a):
    """4x4 generator: Sim3 [tau, phi, sigma] -> [[Phi + sigma I, tau], [0, 0]]; RxSO3 [phi, sigma] without tau."""
    M = np.zeros((4, 4))
    if G is Sim3:
        M[:3, :3] = _hat3(a[3:6]) + a[6] * np.eye(3)
        M[:3, 3] = a[:3]
    else:
        M[:3, :3] = _hat3(a[:3]) + a[3] * np.eye(3)
    return M


def _vee(G, M):
    S = M[:3, :3]
    sigma = np.trace(S) / 3.0
    A = 0.5 * (S - S.T)
    phi = np.array([A[2, 1], A[0, 2], A[1, 0]])
    return np.concatenate([M[:3, 3], phi, [sigma]]) if G is Sim3 else np.concatenate([phi, [sigma]])


def _tangents(G, n, seed, scale=0.6):
    rng = np.random.default_rng(seed)
    a = scale * rng.standard_normal((n, G.manifold_dim))
    # rows exercising every branch of calcW / Exp: tiny rotation, tiny log-scale, both
    if G is Sim3:
        a[0, 3:6] *= 1e-9
        a[1, 6] = 1e-9
        a[2, 3:7] *= 1e-9
    else:
        a[0, :3] *= 1e-9
        a[1, 3] = 1e-9
        a[2] *= 1e-9
    return a


def _devices():
    return [torch.device("cpu")]


def _check_forward(G, dev):
    a = _tangents(G, 12, 0)
    b = _tangents(G, 12, 1, 0.4)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    X, Y = G.exp(T(a)), G.exp(T(b))
    MX = X.matrix().cpu().numpy()
    MY = Y.matrix().cpu().numpy()
    ref = np.stack([scipy.linalg.expm(_hat(G, ai)) for ai in a])
    np.testing.assert_allclose(MX, ref, atol=1e-12)
    # log is the inverse of exp
    np.testing.assert_allclose(X.log().cpu().numpy(), a, atol=1e-9)
    # group law, inverse
    np.testing.assert_allclose((X * Y).matrix().cpu().numpy(), MX @ MY, atol=1e-12)
    np.testing.assert_allclose(X.inv().matrix().cpu().numpy(), np.linalg.inv(MX), atol=1e-11)
    # actions
    rng = np.random.default_rng(5)
    p3, p4 = rng.standard_normal((12, 3)), rng.standard_normal((12, 4))
    q3 = np.einsum("nij,nj->ni", MX[:, :3, :3], p3) + MX[:, :3, 3]
    np.testing.assert_allclose(X.act(T(p3)).cpu().numpy(), q3, atol=1e-12)
    q4 = np.einsum("nij,nj->ni", MX, p4)
    np.testing.assert_allclose(X.act(T(p4)).cpu().numpy(), q4, atol=1e-12)
    # adjoint: hat(Adj_X v) = X hat(v) X^-1; adjT is its transpose
    v, w = rng.standard_normal((12, G.manifold_dim)), rng.standard_normal((12, G.manifold_dim))
    Av = X.adj(T(v)).cpu().numpy()
    for i in range(12):
        np.testing.assert_allclose(_vee(G, MX[i] @ _hat(G, v[i]) @ np.linalg.inv(MX[i])), Av[i], atol=1e-10)
    ATw = X.adjT(T(w)).cpu().numpy()
    np.testing.assert_allclose((ATw * v).sum(-1), (w * Av).sum(-1), atol=1e-10)
    # float32 runs the same closed forms
    X32 = G.exp(T(a).float())
    np.testing.assert_allclose(X32.matrix().cpu().numpy(), ref, atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(X32.log().cpu().numpy()[3:], a[3:], atol=2e-5)


@pytest.mark.parametrize("G", [RxSO3, Sim3], ids=["RxSO3", "Sim3"])
def test_forward_ops_match_matrix_exponential_host(G):
    _check_forward(G, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("G", [RxSO3, Sim3], ids=["RxSO3", "Sim3"])
def test_forward_ops_match_matrix_exponential_device(G):
    _check_forward(G, torch.device("cuda:0"))


def test_sim3_with_unit_scale_is_se3():
    a = torch.from_numpy(_tangents(SE3, 8, 7))
    a7 = torch.cat([a, torch.zeros(8, 1, dtype=a.dtype)], -1)
    X, Z = SE3.exp(a), Sim3.exp(a7)
    assert torch.allclose(Z.data[:, :7], X.data, atol=1e-12) and torch.allclose(Z.data[:, 7], torch.ones(8, dtype=a.dtype))
    assert torch.allclose(Sim3(X).data, Z.data, atol=1e-12)
    assert torch.allclose(Z.log()[:, :6], X.log(), atol=1e-9)
    assert torch.allclose(RxSO3(Z).data, Z.data[:, 3:])


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
        "vec": (lambda a: G.exp(a).vec(), [K]),
    }


def _run_gradcheck(G, name, device):
    fn, dims = _chains(G)[name]
    gen = torch.Generator().manual_seed(11 + len(name))
    scale = 0.02 if G is Sim3 else 0.7
    inputs = []
    for k, d in enumerate(dims):
        s = scale if (k == 0 or name == "mul") else 0.7  # group tangents small for Sim3, vectors / points O(1)
        inputs.append((s * torch.randn(5, d, generator=gen, dtype=torch.float64)).to(device).requires_grad_(True))
    assert torch.autograd.gradcheck(fn, inputs, eps=1e-6, atol=2e-7, rtol=1e-6)


@pytest.mark.parametrize("G", [RxSO3, Sim3], ids=["RxSO3", "Sim3"])
@pytest.mark.parametrize("name", list(_chains(Sim3)))
def test_backward_ops_gradcheck_host(G, name):
    _run_gradcheck(G, name, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("G", [RxSO3, Sim3], ids=["RxSO3", "Sim3"])
@pytest.mark.parametrize("name", list(_chains(Sim3)))
def test_backward_ops_gradcheck_device(G, name):
    _run_gradcheck(G, name, torch.device("cuda:0"))


@pytest.mark.parametrize("G", [RxSO3, Sim3], ids=["RxSO3", "Sim3"])
def test_jinv_is_inverse_left_jacobian_of_log(G):
    gen = torch.Generator().manual_seed(3)
    K = G.manifold_dim
    x = (0.05 if G is Sim3 else 0.6) * torch.randn(4, K, generator=gen, dtype=torch.float64)
    X = G.exp(x)
    a = torch.randn(4, K, generator=gen, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(lambda e: G.exp(e).mul(G(X.data)).log(), torch.zeros(4, K, dtype=torch.float64))
    Jd = torch.stack([J[i, :, i, :] for i in range(4)])
    assert torch.allclose(X.Jinv(a), torch.einsum("nij,nj->ni", Jd, a), atol=1e-7)
