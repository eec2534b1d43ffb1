"""Pin the CPU oracle against outputs of the reference's own Python (tests/golden/*.npz).

The fixtures were produced by tests/golden/make_golden.py in the build container from
/root/reference (loaded by path).  Nothing here reads /root/reference.
"""

import os
import re

import numpy as np
import pytest
import torch

from oracle import ba, corr, geom, se3, update_module
from vipe_amd.synth import make_graph

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _ba_cases():
    src = open(os.path.join(GOLD, "make_golden.py")).read()
    ns = {}
    exec(src[src.index("BA_CASES = {"):src.index("def gen_ba")], ns)
    return ns["BA_CASES"]


BA_CASES = _ba_cases()


def _ba_rig_cases():
    src = open(os.path.join(GOLD, "make_golden.py")).read()
    ns = {}
    exec(src[src.index("BA_RIG_CASES = {"):src.index("def gen_ba_rig")], ns)
    return ns["BA_RIG_CASES"]


BA_RIG_CASES = _ba_rig_cases()


def run_oracle_ba_rig(name, dtype):
    from vipe_amd.synth import make_rig_graph
    gk, bk = BA_RIG_CASES[name]
    g = make_rig_graph(**gk)
    M = g.target.shape[0]
    return ba.bundle_adjustment(g.poses, g.disps, g.disps_sens, g.intrinsics, g.rig, g.target.reshape(M, -1, 2),
                                g.weight.reshape(M, -1, 2), g.eta, g.ii, g.jj, dtype=dtype, **bk), g


@pytest.mark.parametrize("name", sorted(BA_RIG_CASES))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ba_rig_matches_reference_solver(name, dtype):
    """Multi-view rigs (V = 2, 3): per-view intrinsics blocks, the rig-rotation group (view 0 fixed, rotation-only
    retraction, buffer.py:497-506, retractor.py:32-37), cross-view self edges - the oracle against the reference's own
    Solver driven as GraphBuffer.bundle_adjustment drives it (tests/golden/make_golden.py:reference_ba_rig)."""
    G = np.load(os.path.join(GOLD, "ba_rig_reference.npz"))
    (p, d, k, r), g = run_oracle_ba_rig(name, dtype)
    rp, rd, rk, rr = (G[f"{name}/{x}"] for x in ("poses", "disps", "intrinsics", "rig"))
    tol = 2e-5 if dtype == np.float64 else 1e-4
    assert np.abs(p - rp).max() <= tol * max(1.0, np.abs(rp).max())
    assert np.abs(d - rd).max() <= tol * np.abs(rd).max()
    assert np.abs(k - rk).max() <= 1e-4 * np.abs(rk).max()
    assert np.abs(r - rr).max() <= tol
    _, bk = BA_RIG_CASES[name]
    assert np.array_equal(rr[0], g.rig[0])  # view 0 is the gauge
    if bk.get("optimize_rig_rotation"):
        # rotation-only step Exp([0, phi]) * R: the quaternion moves, |t| is preserved (t is rotated about the rig origin)
        assert np.abs(rr[1:, 3:] - g.rig[1:, 3:]).max() > 1e-5
        assert np.allclose(np.linalg.norm(rr[:, :3], axis=1), np.linalg.norm(g.rig[:, :3], axis=1), atol=1e-6)
    else:
        assert np.array_equal(rr, g.rig)
    if bk.get("optimize_intrinsics"):
        assert np.abs(rk[:, :2] - g.intrinsics[:, :2]).min() > 1e-4 and rk[0, 0] != rk[1, 0]  # one block per view
    assert G[name + "/energy"][-1] < G[name + "/energy"][0]


def run_oracle_ba(name, dtype):
    gk, bk = BA_CASES[name]
    bk = dict(bk)
    cam = bk.pop("camera", "pinhole")
    k1 = bk.pop("k1", None)
    g = make_graph(**gk)
    intr = g.intrinsics
    if cam == "mei":
        intr = np.concatenate([intr, np.array([[k1]], dtype=np.float32)], axis=1)
    E = len(g.ii)
    return ba.bundle_adjustment(
        g.poses, g.disps[:, None], g.disps_sens[:, None], intr, se3.se3_identity(1), g.target.reshape(E, -1, 2),
        g.weight.reshape(E, -1, 2), g.eta[:, None], g.ii, g.jj, model=cam, dtype=dtype, **bk), g


@pytest.mark.parametrize("name", sorted(BA_CASES))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ba_matches_reference_solver(name, dtype):
    """north_star tolerance: 1e-4 relative on poses and inverse depth; the oracle sits ~1e-6 from the reference."""
    G = np.load(os.path.join(GOLD, "ba_reference.npz"))
    (p, d, k, _), g = run_oracle_ba(name, dtype)
    rp, rd, rk = G[name + "/poses"], G[name + "/disps"], G[name + "/intrinsics"]
    tol = 2e-5 if "mei_intr" not in name else 1e-4
    assert np.abs(p - rp).max() <= tol * max(1.0, np.abs(rp).max())
    assert np.abs(d[:, 0] - rd).max() <= tol * np.abs(rd).max()
    assert np.abs(k - rk).max() <= 1e-4 * np.abs(rk).max()
    # the step actually moved something (guards against a vacuous comparison)
    assert np.abs(rp - g.poses).max() + np.abs(rd - g.disps).max() > 1e-3


def test_ba_matches_reference_solver_at_the_headline_size():
    """BASELINE configs[2] itself - 48 keyframes, 48 x 64 grid, E = 276, depth prior on, 3 Gauss-Newton iterations, the
    graph `bench.py` times - through the reference `Solver` (`ba_headline_reference.npz`, make_golden.gen_headline)."""
    G = np.load(os.path.join(GOLD, "ba_headline_reference.npz"))
    g = make_graph(n=48, height=384, width=512, radius=3, seed=1234, depth_prior=True)
    E = len(g.ii)
    assert E == int(G["n_edges"][0]) == 276
    p, d, _, _ = ba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics, se3.se3_identity(1),
                                      g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2), g.eta[:, None], g.ii, g.jj,
                                      t0=1, t1=48, n_iters=3, pose_damping=1e-3, pose_ep=0.1)
    rp, rd = G["poses"], G["disps"]
    assert np.abs(p - rp).max() <= 2e-5 * max(1.0, np.abs(rp).max())
    assert np.abs(d[:, 0] - rd).max() <= 2e-5 * np.abs(rd).max()
    assert np.abs(rp - g.poses).max() + np.abs(rd - g.disps).max() > 1e-3 and G["energy"][-1] < G["energy"][0]


def test_ba_energy_decreases():
    G = np.load(os.path.join(GOLD, "ba_reference.npz"))
    for name in BA_CASES:
        e = G[name + "/energy"]
        assert e[-1] < e[0]
    (p, d, k, r), g = run_oracle_ba("n5_frontend", np.float64)
    e0 = ba.energy(g.poses, g.disps[:, None], g.intrinsics, se3.se3_identity(1), g.target, g.weight, g.ii, g.jj)
    e1 = ba.energy(p, d, k, r, g.target, g.weight, g.ii, g.jj)
    assert e1 < e0
    # reference energy at the first iteration equals the oracle's initial energy (solver.py:133-134)
    assert abs(e0 - G["n5_frontend/energy"][0]) < 1e-5 * e0


@pytest.mark.parametrize("cam", ["pinhole", "mei"])
def test_reproject_and_jacobians_match_reference(cam):
    G = np.load(os.path.join(GOLD, "reproject_reference.npz"))
    g = make_graph(n=4, height=64, width=96, radius=2, seed=31)
    intr8 = geom.scaled_intrinsics(G[cam + "/intr"], 1 / 8.0, cam)
    z = np.zeros_like(g.ii)
    for dt, tol in ((np.float32, 1e-6), (np.float64, 2e-6)):
        o = geom.reproject(g.poses.astype(dt), g.disps.astype(dt), intr8.astype(dt), se3.se3_identity(1, dt),
                           g.ii, g.jj, z, z, g.ii, cam, True, True)
        for key in ["coords", "valid", "Ji", "Jj", "Jz", "Jfi", "Jfj"]:
            ref = G[cam + "/" + key]
            assert np.abs(o[key].reshape(ref.shape) - ref).max() <= tol * (np.abs(ref).max() + 1e-30), key


def test_lie_wrapper_semantics():
    """Reference LieGroup wrapper (groups.py) over the oracle backend: retr = Exp(a)*X, adjT broadcasting."""
    G = np.load(os.path.join(GOLD, "lie_wrapper_reference.npz"))
    X, xi, a = G["X"], G["xi"], G["a"]
    assert np.allclose(se3.se3_exp(xi), X, atol=1e-6)
    assert np.allclose(se3.se3_adjT(X[:, None, None], a), G["adjT"], atol=1e-5)
    assert np.allclose(se3.se3_retr(X, xi * np.float32(0.1)), G["retr"], atol=1e-6)
    assert np.allclose(se3.se3_inv(X), G["inv"], atol=1e-6)
    assert np.allclose(se3.se3_log(X), G["log"], atol=1e-5)
    assert np.allclose(se3.se3_log(X), xi, atol=1e-5)
    assert np.allclose(se3.se3_matrix(X), G["matrix"], atol=1e-6)
    ident = se3.se3_mul(X, se3.se3_inv(X))
    assert np.allclose(ident, G["mul"], atol=1e-6)
    assert np.allclose(ident, se3.se3_identity(len(X)), atol=1e-6)


def test_se3_fp64_identities():
    """SE3 closed forms are pinned by group identities in fp64 (no reference binary exists)."""
    rng = np.random.default_rng(0)
    xi = rng.normal(0, 1.0, (1000, 6))
    xi[:, 3:] *= np.minimum(1.0, 3.0 / np.linalg.norm(xi[:, 3:], axis=-1, keepdims=True))  # |phi| < pi
    xi[:10, 3:] *= 1e-9  # theta < EPS Taylor branch (so3.h:139-146)
    xi[10:20, 3:] *= np.pi / np.linalg.norm(xi[10:20, 3:], axis=-1, keepdims=True) * 0.999999  # theta ~ pi
    X = se3.se3_exp(xi)
    assert np.allclose(se3.se3_log(X), xi, atol=1e-6)
    Y = se3.se3_exp(rng.normal(0, 1.0, (1000, 6)))
    p = rng.normal(0, 1, (1000, 4))
    lhs = se3.se3_act4(se3.se3_mul(X, Y), p)
    rhs = se3.se3_act4(X, se3.se3_act4(Y, p))
    assert np.allclose(lhs, rhs, atol=1e-9)
    assert np.allclose(se3.se3_act4(se3.se3_inv(X), se3.se3_act4(X, p)), p, atol=1e-9)
    # Adj(X) a = log(X exp(eps a) X^-1)/eps
    a = rng.normal(0, 1, (1000, 6))
    eps = 1e-6
    conj = se3.se3_mul(se3.se3_mul(X, se3.se3_exp(eps * a)), se3.se3_inv(X))
    assert np.allclose(se3.se3_log(conj) / eps, se3.se3_adj(X, a), atol=1e-4)
    # adjT is the transpose
    b = rng.normal(0, 1, (1000, 6))
    assert np.allclose(np.sum(se3.se3_adjT(X, b) * a, -1), np.sum(b * se3.se3_adj(X, a), -1), atol=1e-9)
    # homogeneous matrix agrees with the action
    T = se3.se3_matrix(X)
    p1 = np.concatenate([p[:, :3], np.ones((1000, 1))], -1)
    assert np.allclose(np.einsum("nij,nj->ni", T, p1), se3.se3_act4(X, p1), atol=1e-9)


def test_corr_pyramid_matches_reference():
    G = np.load(os.path.join(GOLD, "corr_reference.npz"))
    pyr = corr.corr_pyramid(torch.from_numpy(G["fmap1"]), torch.from_numpy(G["fmap2"]))
    for i, lvl in enumerate(pyr):
        assert np.allclose(lvl.numpy(), G[f"level{i}"], atol=1e-4, rtol=1e-5)
    alt = corr.alt_pyramid(torch.from_numpy(G["alt_fmaps"]))
    for i, lvl in enumerate(alt):
        assert np.allclose(lvl.numpy(), G[f"alt_level{i}"], atol=1e-6)


def test_corr_index_against_grid_sample():
    """corr_index_forward == bilinear sampling (zero padding) of each [h2,w2] slab on a 7x7 window,
    output index [i][j] = (x offset, y offset)  (correlation_kernels.cu:50-62)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    B, h1, w1, h2, w2, r = 2, 5, 6, 9, 11, 3
    vol = rng.normal(0, 1, (B, h1, w1, h2, w2)).astype(np.float32)
    coords = np.stack([rng.uniform(-4, w2 + 3, (B, h1, w1)), rng.uniform(-4, h2 + 3, (B, h1, w1))], 1).astype(np.float32)
    out = corr.corr_index_forward(vol, coords, r)
    v = torch.from_numpy(vol).reshape(B * h1 * w1, 1, h2, w2)
    x0 = torch.from_numpy(coords[:, 0]).reshape(-1, 1, 1)
    y0 = torch.from_numpy(coords[:, 1]).reshape(-1, 1, 1)
    off = torch.arange(-r, r + 1).float()
    gx = (x0 + off.view(1, -1, 1)).expand(-1, 7, 7)  # dim1 <-> x offset (i)
    gy = (y0 + off.view(1, 1, -1)).expand(-1, 7, 7)  # dim2 <-> y offset (j)
    grid = torch.stack([2 * gx / (w2 - 1) - 1, 2 * gy / (h2 - 1) - 1], -1)
    ref = F.grid_sample(v, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    ref = ref.reshape(B, h1, w1, 7, 7).permute(0, 3, 4, 1, 2).numpy()
    assert np.allclose(out, ref, atol=2e-5)


def test_corr_index_half_is_half_rounded_chain():
    rng = np.random.default_rng(4)
    vol = rng.normal(0, 1, (1, 3, 4, 8, 8)).astype(np.float16)
    coords = np.stack([rng.uniform(0, 7, (1, 3, 4)), rng.uniform(0, 7, (1, 3, 4))], 1).astype(np.float32)
    o16 = corr.corr_index_forward(vol, coords, 3)
    o32 = corr.corr_index_forward(vol.astype(np.float32), coords, 3)
    assert o16.dtype == np.float16
    assert np.abs(o16.astype(np.float32) - o32).max() < 1e-2
    assert np.abs(o16.astype(np.float32) - o32).max() > 0  # rounding order is observable


def test_altcorr_equals_volume_lookup():
    """altcorr_forward (no volume) == corr_index_forward on the explicit volume (fp32)."""
    rng = np.random.default_rng(5)
    B, H, W, C, r = 2, 4, 8, 64, 3
    f1 = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
    f2 = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
    coords = np.stack([rng.uniform(-2, W + 1, (B, 1, H, W)), rng.uniform(-2, H + 1, (B, 1, H, W))], -1).astype(np.float32)
    a = corr.altcorr_forward(f1, f2, coords, r)  # [B,1,49,H,W]
    vol = np.einsum("bhwc,byxc->bhwyx", f1, f2)
    c = corr.corr_index_forward(vol, np.ascontiguousarray(np.transpose(coords[:, 0], (0, 3, 1, 2))), r)
    assert np.allclose(a[:, 0], c.reshape(B, 49, H, W), atol=1e-4)


def test_update_module_matches_reference():
    from vipe_amd.slam.networks import UpdateModule
    G = np.load(os.path.join(GOLD, "update_module_reference.npz"))
    torch.manual_seed(0)
    um = UpdateModule().eval()
    sd = um.state_dict()
    for k, v in sd.items():
        ref = G["sdsum/" + k]
        assert abs(float(v.double().sum()) - ref[0]) < 1e-6 * max(1, ref[1]), k
    E, ht, wd = 5, 12, 16
    gen = torch.Generator().manual_seed(1)
    net = torch.randn(1, E, 128, ht, wd, generator=gen).tanh()
    inp = torch.randn(1, E, 128, ht, wd, generator=gen).relu()
    cor = torch.randn(1, E, 196, ht, wd, generator=gen)
    flow = torch.randn(1, E, 4, ht, wd, generator=gen) * 4
    for t, s in zip((net, inp, cor, flow), G["input_sums"]):
        assert abs(float(t.double().sum()) - s) < 1e-6 * t.numel()
    ix = torch.from_numpy(G["ix"])
    with torch.no_grad():
        n2, delta, weight, eta, upmask = update_module.update_forward(sd, net, inp, cor, flow, ix)
    assert np.allclose(n2[:, :, ::4].numpy(), G["out_net_sub"], atol=2e-5)
    assert np.allclose(delta.numpy(), G["out_delta"], atol=2e-5)
    assert np.allclose(weight.numpy(), G["out_weight"], atol=2e-5)
    assert np.allclose(eta.numpy(), G["out_eta"], atol=2e-6)
    assert np.allclose(upmask[:, :, ::16].numpy(), G["out_upmask_sub"], atol=2e-5)


def test_update_module_matches_reference_at_the_headline_grid():
    """SURVEY 8(c) golden #1 at [1,4,.,48,64]: the reference class's own outputs (`update_module_headline_reference.npz`)."""
    G = np.load(os.path.join(GOLD, "update_module_headline_reference.npz"))
    from vipe_amd.synth import headline_update_module_inputs
    um, net, inp, cor, flow = headline_update_module_inputs()
    sd = um.state_dict()
    for k, v in sd.items():
        ref = G["sdsum/" + k]
        assert abs(float(v.double().sum()) - ref[0]) < 1e-6 * max(1, ref[1]), k
    for t, s in zip((net, inp, cor, flow), G["input_sums"]):
        assert abs(float(t.double().sum()) - s) < 1e-6 * t.numel()
    with torch.no_grad():
        n2, delta, weight, eta, _ = update_module.update_forward(sd, net, inp, cor, flow, torch.from_numpy(G["ix"]))
    assert np.allclose(n2[:, :, ::8].numpy(), G["out_net_sub"].astype(np.float32), atol=1e-3)  # stored as fp16
    assert np.allclose(delta.numpy(), G["out_delta"], atol=2e-5)
    assert np.allclose(weight.numpy(), G["out_weight"], atol=2e-5)
    assert np.allclose(eta.numpy(), G["out_eta"], atol=2e-6)


@pytest.mark.parametrize("case", ["frontend", "backend", "backend_dense"])
def test_add_proximity_factors_edge_lists_match_reference(case):
    """Golden vector #8: the (ii, jj) lists handed to add_factors by the reference's own
    `add_proximity_factors` (factor_graph.py:411-488, run from the reference file with a fake buffer) - integer-exact,
    including order.  The distances are injected, so this pins the host-side selection / NMS / ordering logic; the
    distances themselves are the `frame_distance` kernel's business (test_gpu_parity)."""
    import types

    from vipe_amd.slam.factor_graph import FactorGraph

    g = np.load(os.path.join(GOLD, "edge_selection_reference.npz"))
    t, t0, t1, rad, nms, maxf = [int(x) for x in g[case + "_params"]]
    beta, thresh = [float(x) for x in g[case + "_beta_thresh"]]
    D = torch.from_numpy(g[case + "_D"])
    fgr = object.__new__(FactorGraph)
    fgr.buffer = types.SimpleNamespace(n_frames=t, n_views=1,
                                       frame_distance_dense_disp=lambda ii, jj, beta: D[ii, jj][:, None])
    fgr.device, fgr.cross_view, fgr.max_factors = torch.device("cpu"), False, maxf
    fgr.ii, fgr.jj = torch.from_numpy(g[case + "_act"][:, 0].copy()), torch.from_numpy(g[case + "_act"][:, 1].copy())
    fgr.ii_inac = torch.from_numpy(g[case + "_inac"][:, 0].copy())
    fgr.jj_inac = torch.from_numpy(g[case + "_inac"][:, 1].copy())
    got = {}
    fgr.add_factors = lambda ii, jj, remove=False: got.update(ii=np.asarray(ii), jj=np.asarray(jj), remove=remove)
    fgr.add_proximity_factors(t0, t1, rad, nms, beta, thresh, True)
    es = g[case + "_es"]
    assert got["remove"] is True
    assert np.array_equal(np.stack([got["ii"], got["jj"]], 1), es)


def _seeded_encoders():
    """fnet, cnet with the weights the fixture was made with (same seed, same construction order as the reference)"""
    from vipe_amd.slam.encoders import BasicEncoder
    G = np.load(os.path.join(GOLD, "encoder_reference.npz"))
    torch.manual_seed(0)
    fnet = BasicEncoder(output_dim=128, norm_fn="instance").eval()
    cnet = BasicEncoder(output_dim=256, norm_fn="none").eval()
    for name, m in (("fnet", fnet), ("cnet", cnet)):
        for k, v in m.state_dict().items():
            ref = G[f"sdsum/{name}.{k}"]
            assert abs(float(v.double().sum()) - ref[0]) < 1e-6 * max(1, ref[1]), (name, k)
    return G, fnet, cnet


def _encoder_images(G, tag, gen):
    V, H, W = [int(x) for x in G[tag + "/shape"]]
    images = torch.rand(V, 3, H, W, generator=gen)
    assert abs(float(images.double().sum()) - float(G[tag + "/image_sum"][0])) < 1e-6 * images.numel()
    return images


def test_encoders_match_reference():
    """oracle/encoder.py against the outputs of the reference's BasicEncoder classes (droid_net.py:290-370) with the
    encode_features / encode_context arithmetic (:510-527)."""
    from oracle import encoder
    G, fnet, cnet = _seeded_encoders()
    gen = torch.Generator().manual_seed(5)
    for tag in ("small", "odd"):
        images = _encoder_images(G, tag, gen)
        with torch.no_grad():
            fmap = encoder.encode_features(fnet.state_dict(), images)
            net, inp = encoder.encode_context(cnet.state_dict(), images)
        assert np.allclose(fmap.numpy(), G[tag + "/fmap"], atol=2e-5)
        assert np.allclose(net.numpy(), G[tag + "/net"], atol=2e-5)
        assert np.allclose(inp.numpy(), G[tag + "/inp"], atol=2e-5)


@pytest.mark.parametrize("name", ["n5_frontend", "n6_window_prior", "n6_infill_motion_limited"])
def test_ba_with_a_sparse_track_term_matches_reference_solver(name):
    """buffer.py:422-447: with sparse tracks enabled the reference adds a SECOND DenseDepthFlowTerm on the same edges.
    The fixture ran both terms through the reference Solver; here the two are folded into one (`fold_flow_terms`: summed
    weights, weighted-mean target - the same normal equations) and given to the single-term oracle."""
    import torch
    from vipe_amd.slam.buffer import fold_flow_terms
    from vipe_amd.synth import make_tracks
    G = np.load(os.path.join(GOLD, "ba_tracks_reference.npz"))
    G1 = np.load(os.path.join(GOLD, "ba_reference.npz"))
    gk, bk = BA_CASES[name]
    g = make_graph(**gk)
    tt, tw = make_tracks(g, 100 + ["n5_frontend", "n6_window_prior", "n6_infill_motion_limited"].index(name))
    E = len(g.ii)
    ft, fw = fold_flow_terms(torch.tensor(g.target.reshape(E, -1, 2)), torch.tensor(g.weight.reshape(E, -1, 2)),
                             torch.tensor(tt.reshape(E, -1, 2)), torch.tensor(tw.reshape(E, -1, 2)))
    p, d, k, _ = ba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics, se3.se3_identity(1),
                                      ft.numpy(), fw.numpy(), g.eta[:, None], g.ii, g.jj, model="pinhole", dtype=np.float64, **bk)
    rp, rd = G[name + "/poses"], G[name + "/disps"]
    assert np.abs(p - rp).max() <= 2e-5 * max(1.0, np.abs(rp).max())
    assert np.abs(d[:, 0] - rd).max() <= 2e-5 * np.abs(rd).max()
    # the track term matters: the two-term result is not the one-term result
    assert np.abs(rp - G1[name + "/poses"]).max() + np.abs(rd - G1[name + "/disps"]).max() > 1e-4


def test_bilinear_splat_matches_reference_function():
    """`oracle.tracks.bilinear_splat` vs outputs of the reference's `bilinear_splatting_inplace` (fixture)."""
    from oracle import tracks
    G = np.load(os.path.join(GOLD, "splat_reference.npz"))
    out, wgt = np.zeros_like(G["out"]), np.zeros_like(G["weight"])
    tracks.bilinear_splat(G["data"], G["uv"], out, wgt)
    assert np.abs(wgt - G["weight"]).max() < 1e-5 and np.abs(out - G["out"]).max() < 1e-4
    assert G["weight"].max() > 1.0  # the reference's corner weights reach 1.5^2: not the textbook bilinear weights


def _synthetic_tracks(seed, n_frames=6, n_kp=60, size=(96, 128)):
    rng = np.random.default_rng(seed)
    pos = rng.uniform([0, 0], [size[1], size[0]], (n_kp, 2)).astype(np.float32)
    vel = rng.normal(0, 1.5, (n_kp, 2)).astype(np.float32)
    obs = []
    for f in range(n_frames):
        seen = rng.random(n_kp) < 0.8
        obs.append({int(k): tuple((pos[k] + f * vel[k]).tolist()) for k in np.flatnonzero(seen)})
    return [obs]


def test_sparse_tracks_target_weight_matches_oracle_on_cpu():
    """`SparseTracks.compute_dense_disp_target_weight` (one upload + two scatter launches for all edges) vs the per-edge
    numpy restatement of sparse_tracks/__init__.py:68-141; correspondences and observations as the reference returns them."""
    import torch
    from oracle import tracks
    from vipe_amd.slam.sparse_tracks import DummySparseTracks, ReplayedSparseTracks
    tr = ReplayedSparseTracks(_synthetic_tracks(8))
    for _ in range(6):
        tr.track_image(None)
    ii = np.array([0, 1, 2, 3, 5, 4, 0]); jj = np.array([1, 0, 4, 3, 2, 5, 5])
    z = torch.zeros(len(ii), dtype=torch.long)
    val, wgt = tr.compute_dense_disp_target_weight(z, torch.tensor(ii), z, torch.tensor(jj), (96, 128), (12, 16))
    rv, rw = tracks.dense_disp_target_weight(tr.observations, [0] * len(ii), ii.tolist(), jj.tolist(), (96, 128), (12, 16))
    assert val.shape == (7, 12, 16, 2) and np.abs(wgt.numpy() - rw).max() < 1e-5 and np.abs(val.numpy() - rv).max() < 1e-4
    assert (rw > 0).sum() > 50 and np.array_equal(rv[3, ..., 0], np.tile(np.arange(16.0), (12, 1)))  # i == j: zero flow
    kp = tr.get_correspondences(0, 0, 1)
    assert kp.tolist() == sorted(set(tr.observations[0][0]) & set(tr.observations[0][1]))
    assert tr.get_observations(0, 1, kp).shape == (len(kp), 2) and tr.get_observations(0, 1, kp[:0]).shape == (0, 2)
    d = DummySparseTracks(2)
    d.track_image(None)
    assert not d.enabled and d.observations == [[{}], [{}]]


def _schedule_fixture():
    import json
    G = np.load(os.path.join(GOLD, "schedule_reference.npz"))
    return G, lambda tag: json.loads(str(G[tag + "/trace"]))


def _as_lists(trace):
    import json
    return json.loads(json.dumps(trace))


@pytest.mark.parametrize("tag,has_pose,seq_init", [("frontend", False, True), ("frontend_init_pose_no_seq", True, False)])
def test_frontend_schedule_matches_the_reference_class(tag, has_pose, seq_init, monkeypatch):
    """`SLAMFrontend` against the reference's own class (frontend.py:32-167, run in the build container with the recording
    fakes of tests/golden/fakes.py): the exact sequence of graph / buffer calls over 8 warm-up + 9 further keyframes - two of
    which fail the keyframe-distance test and are dropped again - with every argument (window bounds, radii, thresholds,
    masks of aged edges, `use_inactive`, `fixed_motion`, `t0`), the extrapolated poses (Exp(0.5 Log(.)) through lietorch) and
    the next-frame disparities."""
    sys_path = os.path.join(GOLD)
    import sys
    sys.path.insert(0, sys_path)
    import fakes
    from vipe_amd.slam import frontend as fe
    G, trace_of = _schedule_fixture()
    monkeypatch.setattr(fe, "FactorGraph", fakes.FakeGraph)
    args = fe.FrontendArgs(warmup=8, beta=0.3, keyframe_thresh=4.0, frontend_thresh=16.0, frontend_window=25, frontend_radius=2,
                           frontend_nms=1, seq_init=seq_init, has_init_pose=has_pose, cross_view=True)
    trace, poses, disps, t1, n = fakes.run_frontend(fe.SLAMFrontend, args, has_pose)
    want = trace_of(tag)
    got = _as_lists(trace)
    assert len(got) == len(want)
    for k, (a, b) in enumerate(zip(got, want)):
        assert a == b, (k, a, b)
    assert [t1, n] == G[tag + "/t1_n"].tolist()
    assert np.abs(poses.numpy() - G[tag + "/poses"]).max() < 1e-6
    assert np.abs(disps.numpy() - G[tag + "/disps"]).max() < 1e-6


@pytest.mark.parametrize("tag,n_frames,depth,opt_intr,adaptive,edges", [("backend", 12, False, False, False, 3),
                                                                       ("backend_depth_intr", 12, True, True, True, 3),
                                                                       ("backend_single", 1, False, False, False, 0)])
def test_backend_schedule_matches_the_reference_class(tag, n_frames, depth, opt_intr, adaptive, edges, monkeypatch):
    """`SLAMBackend.run` / `run_if_necessary` against the reference's class (backend.py:31-122): graph construction
    (max_factors = 16 t, non-incremental), proximity arguments, the plain / depth-prior (half the passes, sensor disparities
    refreshed, the rest with the intrinsics held) / single-keyframe branches, 16 instead of 8 GN iterations when intrinsics
    are optimised."""
    import sys
    sys.path.insert(0, GOLD)
    import fakes
    from vipe_amd.slam import backend as be
    G, trace_of = _schedule_fixture()
    monkeypatch.setattr(be, "FactorGraph", fakes.FakeGraph)
    monkeypatch.setattr(fakes.FakeGraph, "edges_per_add", edges)
    video = fakes.FakeBuffer(n_frames)
    video.disps_sens[0, 0, 0] = 0.7
    args = be.BackendArgs(beta=0.3, backend_thresh=22.0, backend_radius=2, backend_nms=3, optimize_intrinsics=opt_intr,
                          optimize_rig_rotation=False, cross_view=True, adaptive_cross_view=adaptive)
    b = be.SLAMBackend(None, video, args, torch.device("cpu"))
    b.depth_model = object() if depth else None
    b.run(7)
    b.run_if_necessary(5)
    assert _as_lists(video.trace) == trace_of(tag)
    assert np.abs(video.disps[0].numpy() - G[tag + "/disps0"]).max() == 0


@pytest.mark.parametrize("tag,dense", [("infill", False), ("infill_dense_disp", True)])
def test_inner_filler_matches_the_reference_class(tag, dense, monkeypatch):
    """`InnerFiller` against the reference's class (inner_filler.py:46-138): for 12 frames appended behind keyframes at
    frames 0, 3, 6, 9 in two chunks - the neighbouring keyframes of every frame (the edges handed to `add_factors`), the
    constant-velocity poses, the ten `update` calls with their flags, the disparity initialisation of `infill_dense_disp`."""
    import sys
    sys.path.insert(0, GOLD)
    import fakes
    from vipe_amd.slam import inner_filler as inf
    G, trace_of = _schedule_fixture()
    monkeypatch.setattr(inf, "FactorGraph", fakes.FakeGraph)
    video = fakes.FakeBuffer(4, seed=3)
    video.tstamp[:4] = torch.tensor([0, 3, 6, 9])
    video.disps_sens[4:10, 0, 0] = 0.9
    f = inf.InnerFiller(None, video, inf.InfillArgs(infill_chunk_size=6, infill_dense_disp=dense), torch.device("cpu"))
    f.set_start_idx(4)
    for frame in range(12):
        video.tstamp[video.n_frames] = frame
        video.n_frames += 1
        if f.check() or frame == 11:
            f.compute()
    res = f.get_result()
    assert _as_lists(video.trace) == trace_of(tag)
    # frames beyond the last keyframe divide log(G G^-1) - rounding noise around zero - by d_time = 1e-3 and multiply by the
    # frame distance (inner_filler.py:73-77): the noise of two correct fp32 implementations differs by a few 1e-5 there
    assert np.abs(res.poses.data.numpy() - G[tag + "/filled_poses"]).max() < 1e-4
    inside = slice(0, 10)  # frames 0..9 lie between keyframes
    assert np.abs(res.poses.data.numpy()[inside] - G[tag + "/filled_poses"][inside]).max() < 2e-6
    if dense:
        assert np.abs(res.dense_disps.numpy() - G[tag + "/filled_disps"]).max() < 1e-6
