"""The PyTorch-CPU corr + BA path that bench.py times as `cpu_baseline` (oracle/torch_cpu.py) against the numpy oracle
(oracle/corr.py, oracle/ba.py - themselves pinned to the reference's Python by tests/test_oracle_golden.py)."""
import numpy as np
import torch

from oracle import ba as oba
from oracle import corr as ocorr
from oracle import se3 as ose3
from oracle import torch_cpu as tc
from vipe_amd.synth import make_graph


def test_pyramid_and_lookup_match_numpy_oracle():
    torch.manual_seed(0)
    E, C, h, w = 3, 16, 8, 16
    f1, f2 = torch.randn(E, C, h, w), torch.randn(E, C, h, w)
    pyr = tc.corr_pyramid(f1, f2)
    ref = ocorr.corr_pyramid(f1[None], f2[None])
    for a, b in zip(pyr, ref):
        assert torch.allclose(a, b, atol=1e-5)
    rng = np.random.default_rng(1)
    coords = np.stack([rng.uniform(-3, w + 3, (E, h, w)), rng.uniform(-3, h + 3, (E, h, w))], -1).astype(np.float32)
    got = tc.corr_lookup(pyr, torch.from_numpy(coords)).numpy()
    want = ocorr.corr_lookup([p.numpy() for p in pyr], coords[None])[0]
    assert got.shape == want.shape == (E, 196, h, w)
    assert np.abs(got - want).max() < 2e-5


def test_dense_ba_matches_numpy_oracle():
    for prior in (False, True):
        g = make_graph(n=6, height=96, width=128, radius=2, seed=3, depth_prior=prior)
        E = len(g.ii)
        kw = dict(t0=1, t1=6, n_iters=3, pose_damping=1e-3, pose_ep=0.1)
        T = torch.from_numpy
        p, d = tc.bundle_adjustment(T(g.poses), T(g.disps), T(g.disps_sens), T(g.intrinsics), T(g.target.reshape(E, -1, 2)),
                                    T(g.weight.reshape(E, -1, 2)), T(g.eta), T(g.ii), T(g.jj), **kw)
        op, od, _, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                             ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                             g.eta[:, None], g.ii, g.jj, **kw)
        assert np.abs(p.numpy() - op).max() < 2e-4, np.abs(p.numpy() - op).max()
        assert np.abs(d.numpy() - od[:, 0]).max() / np.abs(od).max() < 2e-4
        assert np.abs(p.numpy() - g.poses).max() > 1e-4  # it moved
