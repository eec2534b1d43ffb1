"""Seeded synthetic factor graphs for tests and bench (no video, no weights).

Shapes follow SURVEY.md section 8(d): N keyframes on a smooth trajectory, pinhole
fx = fy = 0.9 * width, disparities 1/U(1,5), radius-r bidirectional neighbourhood
edges (what add_proximity_factors(rad=2) always inserts, factor_graph.py:467-470),
targets = reprojection of the ground truth + N(0, 0.5 px), weights U(0,1),
eta = 0.01 * softplus(N(0,1)).  Pure numpy so it runs without a GPU.
"""

from dataclasses import dataclass

import numpy as np


def _qmul(a, b):
    ax, ay, az, aw = np.moveaxis(a, -1, 0)
    bx, by, bz, bw = np.moveaxis(b, -1, 0)
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], -1)


def _qrot(q, p):
    qv, w = q[..., :3], q[..., 3:4]
    uv = 2 * np.cross(qv, p)
    return p + w * uv + np.cross(qv, uv)


def _qexp(phi):
    th = np.linalg.norm(phi, axis=-1, keepdims=True)
    th = np.maximum(th, 1e-12)
    return np.concatenate([np.sin(0.5 * th) / th * phi, np.cos(0.5 * th)], -1)


def _se3_mul(A, B):
    return np.concatenate([A[..., :3] + _qrot(A[..., 3:], B[..., :3]), _qmul(A[..., 3:], B[..., 3:])], -1)


def _se3_inv(A):
    qi = A[..., 3:] * np.array([-1, -1, -1, 1.0])
    return np.concatenate([-_qrot(qi, A[..., :3]), qi], -1)


def neighbourhood_edges(n, radius=3):
    """All ordered pairs 0 < |i-j| <= radius (factor_graph.py:396-409); 276 edges at n=48, r=3."""
    ii, jj = [], []
    for i in range(n):
        for j in range(max(i - radius, 0), i):
            ii += [i, j]
            jj += [j, i]
    return np.asarray(ii, dtype=np.int64), np.asarray(jj, dtype=np.int64)


@dataclass
class SyntheticGraph:
    ht: int
    wd: int
    n: int
    poses_gt: np.ndarray  # [n,7] world->camera
    disps_gt: np.ndarray  # [n,ht,wd]
    poses: np.ndarray  # perturbed initial estimate
    disps: np.ndarray
    disps_sens: np.ndarray
    intrinsics: np.ndarray  # [1,4] FULL resolution (divided by 8 inside the path)
    ii: np.ndarray
    jj: np.ndarray
    target: np.ndarray  # [E,ht,wd,2]
    weight: np.ndarray  # [E,ht,wd,2]
    eta: np.ndarray  # [n,ht,wd]


def reproject_pinhole(poses, disps, intr8, ii, jj):
    """coords [E,ht,wd,2] of pixels of frame ii seen in frame jj (float64)."""
    n, ht, wd = disps.shape
    v, u = np.meshgrid(np.arange(ht, dtype=np.float64), np.arange(wd, dtype=np.float64), indexing="ij")
    fx, fy, cx, cy = intr8
    X0 = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
    T = _se3_mul(poses[jj], _se3_inv(poses[ii]))
    d = disps[ii]
    X1 = _qrot(T[:, None, None, 3:], np.broadcast_to(X0, d.shape + (3,))) + T[:, None, None, :3] * d[..., None]
    Z = np.where(X1[..., 2] < 0.1, 1.0, X1[..., 2])
    return np.stack([fx * X1[..., 0] / Z + cx, fy * X1[..., 1] / Z + cy], -1)


def expand_edges(ii, jj, V, cross_view_idx=None):
    """buffer.py:318-361 on the host: every edge becomes V terms (pi,qi) -> (pj,qj), same view on both ends, except
    self edges (i == j), which pair view v with view cross_view_idx[v] of the same keyframe.  -> pi, qi, di, pj, qj"""
    ii, jj = np.asarray(ii, np.int64), np.asarray(jj, np.int64)
    cv = np.asarray(cross_view_idx if cross_view_idx is not None else [(v + 1) % V for v in range(V)], np.int64)
    pi, pj = np.repeat(ii, V), np.repeat(jj, V)
    qi = np.tile(np.arange(V, dtype=np.int64), len(ii))
    qj = np.where(pi == pj, cv[qi], qi)
    return pi, qi, pi * V + qi, pj, qj


@dataclass
class SyntheticRigGraph:
    ht: int
    wd: int
    n: int
    V: int
    poses_gt: np.ndarray  # [n,7] world->rig
    rig_gt: np.ndarray  # [V,7] camera v -> rig (view 0 = identity)
    disps_gt: np.ndarray  # [n,V,ht,wd]
    poses: np.ndarray
    rig: np.ndarray  # perturbed rotations for views >= 1
    disps: np.ndarray
    disps_sens: np.ndarray
    intrinsics: np.ndarray  # [V,4] full resolution
    ii: np.ndarray  # [E] keyframe edges incl. self edges (i,i) = cross-view terms
    jj: np.ndarray
    target: np.ndarray  # [E*V,ht,wd,2]
    weight: np.ndarray
    eta: np.ndarray  # [n,V,ht,wd]


def make_rig_graph(n=4, V=2, height=96, width=128, radius=2, seed=3, self_edges=True, depth_prior=False,
                   pose_noise=0.003, disp_noise=0.05, rig_noise=0.004):
    """A small multi-camera rig clip (n keyframes x V views): view v looks ~0.06 v rad to the side of view 0 from a
    0.08 v baseline, per-view focal lengths differ by a few percent.  Edges: the radius-r neighbourhood between
    keyframes plus, with `self_edges`, (i,i) for every keyframe - which `expand_edges` turns into the cross-view terms
    (factor_graph.py:62,454-458).  Targets = ground-truth reprojection + N(0, 0.5 px)."""
    rng = np.random.default_rng(seed)
    ht, wd = height // 8, width // 8
    k = np.arange(n, dtype=np.float64)
    c2w = np.concatenate([np.stack([0.05 * k, np.zeros(n), np.zeros(n)], -1) + rng.normal(0, 0.01, (n, 3)),
                          _qexp(rng.normal(0, 0.01, (n, 3)))], -1)
    c2w[0] = [0, 0, 0, 0, 0, 0, 1]
    poses_gt = _se3_inv(c2w)
    rig_gt = np.zeros((V, 7))
    rig_gt[:, 6] = 1
    for v in range(1, V):
        rig_gt[v] = np.concatenate([[0.08 * v, 0.01 * v, 0.0], _qexp(np.array([0.0, 0.06 * v, 0.01 * v]))])
    disps_gt = 1.0 / rng.uniform(1.0, 5.0, (n, V, ht, wd))
    intr = np.stack([[0.9 * width * (1 + 0.03 * v), 0.9 * width * (1 + 0.03 * v), width / 2.0 + v, height / 2.0 - v]
                     for v in range(V)])
    ii, jj = neighbourhood_edges(n, radius)
    if self_edges:
        ii, jj = np.concatenate([ii, np.arange(n)]), np.concatenate([jj, np.arange(n)])
    pi, qi, di, pj, qj = expand_edges(ii, jj, V)
    M = len(pi)

    def reproject(poses, rig, disps):
        T = _se3_mul(_se3_mul(_se3_inv(rig[qj]), _se3_mul(poses[pj], _se3_inv(poses[pi]))), rig[qi])
        v_, u_ = np.meshgrid(np.arange(ht, dtype=np.float64), np.arange(wd, dtype=np.float64), indexing="ij")
        Ii, Ij = intr[qi] / 8.0, intr[qj] / 8.0
        X0 = np.stack([(u_[None] - Ii[:, 2, None, None]) / Ii[:, 0, None, None],
                       (v_[None] - Ii[:, 3, None, None]) / Ii[:, 1, None, None], np.ones((M, ht, wd))], -1)
        d = disps.reshape(n * V, ht, wd)[di]
        X1 = _qrot(T[:, None, None, 3:], X0) + T[:, None, None, :3] * d[..., None]
        Z = np.where(X1[..., 2] < 0.1, 1.0, X1[..., 2])
        return np.stack([Ij[:, 0, None, None] * X1[..., 0] / Z + Ij[:, 2, None, None],
                         Ij[:, 1, None, None] * X1[..., 1] / Z + Ij[:, 3, None, None]], -1)

    target = reproject(poses_gt, rig_gt, disps_gt) + rng.normal(0, 0.5, (M, ht, wd, 2))
    weight = rng.uniform(0, 1, (M, ht, wd, 2))
    eta = 0.01 * np.log1p(np.exp(rng.normal(0, 1, (n, V, ht, wd))))
    dT = np.concatenate([rng.normal(0, pose_noise, (n, 3)), _qexp(rng.normal(0, pose_noise, (n, 3)))], -1)
    dT[0] = [0, 0, 0, 0, 0, 0, 1]
    poses = _se3_mul(dT, poses_gt)
    rig = rig_gt.copy()
    for v in range(1, V):  # rotation-only perturbation: what optimize_rig_rotation can undo
        rig[v] = _se3_mul(np.concatenate([[0, 0, 0], _qexp(rng.normal(0, rig_noise, 3))]), rig_gt[v])
    disps = disps_gt * (1 + rng.normal(0, disp_noise, disps_gt.shape))
    sens = disps_gt * (1 + rng.normal(0, 0.05, disps_gt.shape)) if depth_prior else np.zeros_like(disps_gt)
    f32 = np.float32
    return SyntheticRigGraph(ht, wd, n, V, poses_gt.astype(f32), rig_gt.astype(f32), disps_gt.astype(f32),
                             poses.astype(f32), rig.astype(f32), disps.astype(f32), sens.astype(f32), intr.astype(f32),
                             ii, jj, target.astype(f32), weight.astype(f32), eta.astype(f32))


def make_graph(n=48, height=384, width=512, radius=3, extra_edges=0, seed=1234, depth_prior=False,
               pose_noise=0.003, disp_noise=0.05):
    """Config 2/3 of BASELINE.json at the defaults (E=276); config 1 at n=2, 96x128."""
    rng = np.random.default_rng(seed)
    ht, wd = height // 8, width // 8
    k = np.arange(n, dtype=np.float64)
    c2w_t = np.stack([0.05 * k, np.zeros(n), np.zeros(n)], -1) + rng.normal(0, 0.01, (n, 3))
    c2w_q = _qexp(rng.normal(0, 0.01, (n, 3)))
    c2w = np.concatenate([c2w_t, c2w_q], -1)
    c2w[0] = [0, 0, 0, 0, 0, 0, 1]
    poses_gt = _se3_inv(c2w)
    disps_gt = 1.0 / rng.uniform(1.0, 5.0, (n, ht, wd))
    intr = np.array([[0.9 * width, 0.9 * width, width / 2.0, height / 2.0]])
    ii, jj = neighbourhood_edges(n, radius)
    if extra_edges > 0:
        have = set(zip(ii.tolist(), jj.tolist()))
        ei, ej = [], []
        while len(ei) < extra_edges:
            a, b = rng.integers(0, n, 2)
            if a != b and (a, b) not in have:
                have.add((int(a), int(b)))
                ei.append(int(a))
                ej.append(int(b))
        ii = np.concatenate([ii, np.asarray(ei, dtype=np.int64)])
        jj = np.concatenate([jj, np.asarray(ej, dtype=np.int64)])
    E = len(ii)
    target = reproject_pinhole(poses_gt, disps_gt, intr[0] / 8.0, ii, jj) + rng.normal(0, 0.5, (E, ht, wd, 2))
    weight = rng.uniform(0, 1, (E, ht, wd, 2))
    eta = 0.01 * np.log1p(np.exp(rng.normal(0, 1, (n, ht, wd))))
    # perturbed starting point (pose 0 stays the gauge)
    dq = _qexp(rng.normal(0, pose_noise, (n, 3)))
    dT = np.concatenate([rng.normal(0, pose_noise, (n, 3)), dq], -1)
    dT[0] = [0, 0, 0, 0, 0, 0, 1]
    poses = _se3_mul(dT, poses_gt)
    poses[:, 3:] /= np.linalg.norm(poses[:, 3:], axis=-1, keepdims=True)
    disps = disps_gt * (1 + rng.normal(0, disp_noise, disps_gt.shape))
    sens = disps_gt * (1 + rng.normal(0, 0.05, disps_gt.shape)) if depth_prior else np.zeros_like(disps_gt)
    f32 = np.float32
    return SyntheticGraph(ht, wd, n, poses_gt.astype(f32), disps_gt.astype(f32), poses.astype(f32),
                          disps.astype(f32), sens.astype(f32), intr.astype(f32), ii, jj,
                          target.astype(f32), weight.astype(f32), eta.astype(f32))


def make_tracks(g, seed):
    """A second (target, weight) pair on the graph's edges, shaped like what `SparseTracks.compute_dense_disp_target_weight`
    returns: about 4 % of the pixels carry a track observation (the flow target displaced by up to a pixel, weights of
    the order of the dense ones), everything else has weight 0."""
    rng = np.random.default_rng(seed)
    target = (g.target + rng.normal(0, 0.7, g.target.shape)).astype(np.float32)
    hit = rng.random(g.weight.shape[:-1] + (1,)) < 0.04
    weight = (hit * rng.uniform(0.5, 3.0, g.weight.shape)).astype(np.float32)
    return target, weight


def headline_update_module_inputs():
    """the inputs / weights `make_golden.gen_headline` fed the reference UpdateModule (re-created from the seeds)"""
    import torch

    from .slam.networks import UpdateModule
    torch.manual_seed(0)
    um = UpdateModule().eval()
    E, ht, wd = 4, 48, 64
    gen = torch.Generator().manual_seed(21)
    net = torch.randn(1, E, 128, ht, wd, generator=gen).tanh()
    inp = torch.randn(1, E, 128, ht, wd, generator=gen).relu()
    cor = torch.randn(1, E, 196, ht, wd, generator=gen)
    flow = torch.randn(1, E, 4, ht, wd, generator=gen) * 4
    return um, net, inp, cor, flow
