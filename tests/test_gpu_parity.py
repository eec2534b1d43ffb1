"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Run on the MI355X box with `pytest -m gpu`.  Tolerances: bit-exact for fp16 correlation lookups and all
index work; 1e-4 relative on poses / inverse depth (BASELINE.json north_star), tighter where noted.
"""

import os

import numpy as np
import pytest
import torch

from oracle import ba as oba
from oracle import corr as ocorr
from oracle import geom as ogeom
from oracle import se3 as ose3
from vipe_amd.synth import make_graph

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def T(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev())


# ------------------------------------------------------------------------------------------------ correlation


@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64])
def test_corr_index_forward(dtype):
    from vipe_amd.ext import droid_net_ext
    rng = np.random.default_rng(0)
    B, h1, w1, h2, w2, r = 3, 6, 8, 12, 16, 3
    vol = rng.normal(0, 2, (B, h1, w1, h2, w2)).astype(dtype)
    coords = np.stack([rng.uniform(-5, w2 + 4, (B, h1, w1)), rng.uniform(-5, h2 + 4, (B, h1, w1))], 1).astype(np.float32)
    coords[0, :, 0, 0] = [3.0, 5.0]  # integer coordinates: dx = dy = 0 (identity-flow case)
    ref = ocorr.corr_index_forward(vol, coords, r)
    (out,) = droid_net_ext.corr_index_forward(T(vol), T(coords), r)
    out = out.cpu().numpy()
    assert out.dtype == dtype and out.shape == ref.shape
    if dtype == np.float16:
        assert np.array_equal(out.view(np.uint16), ref.view(np.uint16)), "fp16 lookup must be bit-exact"
    else:
        # float/double use a fused multiply-add per tap (nvcc contraction); the numpy oracle rounds twice
        tol = 2e-6 if dtype == np.float32 else 1e-14
        assert np.abs(out - ref).max() <= tol * np.abs(ref).max()


def test_corr_index_forward_ragged_and_empty():
    from vipe_amd.ext import droid_net_ext
    rng = np.random.default_rng(1)
    # non-multiple-of-tile sizes, window entirely outside the map
    vol = rng.normal(0, 1, (1, 5, 7, 3, 5)).astype(np.float16)
    coords = np.full((1, 2, 5, 7), -100.0, dtype=np.float32)
    (out,) = droid_net_ext.corr_index_forward(T(vol), T(coords), 3)
    assert torch.count_nonzero(out) == 0
    (empty,) = droid_net_ext.corr_index_forward(T(vol[:0]), T(coords[:0]), 3)
    assert empty.shape == (0, 7, 7, 5, 7)
    with pytest.raises(RuntimeError):
        droid_net_ext.corr_index_forward(T(vol).permute(0, 2, 1, 3, 4), T(coords), 3)  # non-contiguous


def test_corr_pyramid_lookup_matches_per_level_and_oracle():
    from vipe_amd.ext import droid_net_ext
    from vipe_amd.slam.networks import CorrBlock
    rng = np.random.default_rng(2)
    E, C, h, w = 3, 128, 16, 24
    f1 = torch.from_numpy(rng.normal(0, 1, (1, E, C, h, w)).astype(np.float16)).to(dev())
    f2 = torch.from_numpy(rng.normal(0, 1, (1, E, C, h, w)).astype(np.float16)).to(dev())
    cb = CorrBlock(f1, f2)
    coords = np.stack([rng.uniform(-2, w + 1, (1, E, h, w)), rng.uniform(-2, h + 1, (1, E, h, w))], -1).astype(np.float32)
    out = cb(T(coords))
    assert out.shape == (1, E, 196, h, w) and out.dtype == torch.float16
    # per-level reference entry point on the same pyramid
    c2 = T(np.ascontiguousarray(np.transpose(coords[0], (0, 3, 1, 2))))
    per = []
    for i, lvl in enumerate(cb.corr_pyramid):
        (o,) = droid_net_ext.corr_index_forward(lvl.contiguous(), (c2 / 2**i).contiguous(), 3)
        per.append(o.view(E, 49, h, w))
    per = torch.cat(per, 1)
    assert torch.equal(out[0], per)
    # oracle on the device-built pyramid: bit-exact
    ref = ocorr.corr_lookup([lv.cpu().numpy() for lv in cb.corr_pyramid], coords, 3)
    assert np.array_equal(out.cpu().numpy().view(np.uint16), ref.view(np.uint16))


def test_corr_index_backward_is_adjoint():
    from vipe_amd.ext import droid_net_ext
    rng = np.random.default_rng(3)
    B, h1, w1, h2, w2, r = 2, 4, 5, 9, 10, 3
    vol = rng.normal(0, 1, (B, h1, w1, h2, w2)).astype(np.float32)
    coords = np.stack([rng.uniform(-2, w2 + 1, (B, h1, w1)), rng.uniform(-2, h2 + 1, (B, h1, w1))], 1).astype(np.float32)
    g = rng.normal(0, 1, (B, 7, 7, h1, w1)).astype(np.float32)
    (out,) = droid_net_ext.corr_index_forward(T(vol), T(coords), r)
    (gv,) = droid_net_ext.corr_index_backward(T(vol), T(coords), T(g), r)
    lhs = float((out.double() * T(g).double()).sum())
    rhs = float((gv.double() * T(vol).double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
    ref = ocorr.corr_index_backward(vol.shape, coords, g, r)
    assert np.abs(gv.cpu().numpy() - ref).max() <= 1e-5


# ------------------------------------------------------------------------------------------------ lietorch


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_lietorch_device_ops_match_oracle(dtype):
    from vipe_amd.ext.lietorch import SE3
    rng = np.random.default_rng(4)
    n = 1000
    npdt = np.float32 if dtype == torch.float32 else np.float64
    xi = rng.normal(0, 1, (n, 6)).astype(npdt)
    xi[:8, 3:] *= 1e-8  # Taylor branch
    a = rng.normal(0, 1, (n, 6)).astype(npdt)
    p = rng.normal(0, 1, (n, 4)).astype(npdt)
    tol = 2e-5 if dtype == torch.float32 else 1e-11
    X = SE3.exp(T(xi))
    Xo = ose3.se3_exp(xi.astype(np.float64))
    assert np.abs(X.data.cpu().numpy() - Xo).max() < tol
    Y = SE3.exp(T(a * 0.3))
    Yo = ose3.se3_exp((a * 0.3).astype(np.float64))
    assert np.abs((X * Y).data.cpu().numpy() - ose3.se3_mul(Xo, Yo)).max() < tol
    assert np.abs(X.inv().data.cpu().numpy() - ose3.se3_inv(Xo)).max() < tol
    assert np.abs(X.log().cpu().numpy() - ose3.se3_log(Xo)).max() < 20 * tol
    assert np.abs(X.act(T(p)).cpu().numpy() - ose3.se3_act4(Xo, p.astype(np.float64))).max() < tol
    assert np.abs(X.act(T(p[:, :3].copy())).cpu().numpy() - ose3.se3_act3(Xo, p[:, :3].astype(np.float64))).max() < tol
    assert np.abs(X.adj(T(a)).cpu().numpy() - ose3.se3_adj(Xo, a.astype(np.float64))).max() < 5 * tol
    assert np.abs(X.adjT(T(a)).cpu().numpy() - ose3.se3_adjT(Xo, a.astype(np.float64))).max() < 5 * tol
    assert np.abs(X.matrix().cpu().numpy() - ose3.se3_matrix(Xo)).max() < tol
    assert np.abs(X.retr(T(a * 0.1)).data.cpu().numpy() - ose3.se3_retr(Xo, (a * 0.1).astype(np.float64))).max() < tol
    # device path == host path of the same library (same closed forms)
    Xc = SE3.exp(torch.from_numpy(xi))
    assert np.abs(Xc.data.numpy() - X.data.cpu().numpy()).max() < tol


def test_lietorch_broadcast_adjT_act4():
    from vipe_amd.ext.lietorch import SE3
    G = np.load(os.path.join(GOLD, "lie_wrapper_reference.npz"))
    X = SE3(T(G["X"]))
    out = X[:, None, None].adjT(T(G["a"]))
    assert np.abs(out.cpu().numpy() - G["adjT"]).max() < 1e-5
    p = torch.randn(6, 5, 3, 4, device=dev())
    full = SE3(X.data[:, None, None].expand(6, 5, 3, 7).contiguous()).act(p)
    assert torch.allclose(X[:, None, None].act(p), full, atol=1e-6)


# ------------------------------------------------------------------------------------------------ reprojection


@pytest.mark.parametrize("cam", ["pinhole", "mei"])
def test_reproject_matches_reference_fixture(cam):
    from vipe_amd.ext import slam_ext
    G = np.load(os.path.join(GOLD, "reproject_reference.npz"))
    g = make_graph(n=4, height=64, width=96, radius=2, seed=31)
    z = np.zeros_like(g.ii)
    rig = ose3.se3_identity(1)
    coords, valid = slam_ext.reproject(T(g.poses), T(g.disps), T(G[cam + "/intr"]), T(rig), T(g.ii), T(z), T(g.jj), T(z),
                                       T(g.ii), camera=cam)
    ref = G[cam + "/coords"]
    assert np.abs(coords.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    assert np.array_equal(valid.cpu().numpy(), G[cam + "/valid"])


def test_reproject_motion_fused():
    from vipe_amd.ext import slam_ext
    g = make_graph(n=5, height=96, width=128, radius=2, seed=3)
    z = np.zeros_like(g.ii)
    args = (T(g.poses), T(g.disps), T(g.intrinsics), T(ose3.se3_identity(1)), T(g.ii), T(z), T(g.jj), T(z), T(g.ii))
    coords, _ = slam_ext.reproject(*args)
    c2, motn = slam_ext.reproject_motion(*args, T(g.target), motn_dtype=torch.float32)
    assert torch.equal(coords, c2)
    ht, wd = g.ht, g.wd
    yy, xx = torch.meshgrid(torch.arange(ht, device=dev()).float(), torch.arange(wd, device=dev()).float(), indexing="ij")
    grid = torch.stack([xx, yy], -1)
    ref = torch.cat([coords - grid, T(g.target) - coords], -1).permute(0, 3, 1, 2).clamp(-64, 64)
    assert torch.equal(motn, ref.contiguous())


# ------------------------------------------------------------------------------------------------ dense BA


def _ba_cases():
    src = open(os.path.join(GOLD, "make_golden.py")).read()
    ns = {}
    exec(src[src.index("BA_CASES = {"):src.index("def gen_ba")], ns)
    return ns["BA_CASES"]


BA_CASES = _ba_cases()


def run_hip_ba(g, intr, cam, bk, n_views=1, rig=None):
    from vipe_amd.ext import slam_ext
    E = len(g.ii)
    poses = T(g.poses).clone()
    disps = T(g.disps).clone()
    intr_t = T(intr).clone()
    rig_t = T(ose3.se3_identity(1) if rig is None else rig)
    z = np.zeros_like(g.ii)
    info = slam_ext.dense_ba(poses, disps, T(g.disps_sens), intr_t, rig_t, T(g.target.reshape(E, -1, 2)),
                             T(g.weight.reshape(E, -1, 2)), T(g.eta), T(g.ii), T(z), T(g.jj), T(z), T(g.ii),
                             camera=cam, want_info=True, **bk)
    torch.cuda.synchronize()
    return poses.cpu().numpy(), disps.cpu().numpy(), intr_t.cpu().numpy(), info.cpu().numpy()


@pytest.mark.parametrize("name", sorted(BA_CASES))
def test_dense_ba_matches_reference_solver(name):
    """HIP BA vs the output of the reference's own Python Solver (fixture), 1e-4 relative (north_star)."""
    G = np.load(os.path.join(GOLD, "ba_reference.npz"))
    gk, bk = BA_CASES[name]
    bk = dict(bk)
    cam = bk.pop("camera", "pinhole")
    k1 = bk.pop("k1", None)
    g = make_graph(**gk)
    intr = g.intrinsics if cam == "pinhole" else np.concatenate([g.intrinsics, np.array([[k1]], np.float32)], 1)
    p, d, k, info = run_hip_ba(g, intr, cam, bk)
    rp, rd, rk = G[name + "/poses"], G[name + "/disps"], G[name + "/intrinsics"]
    assert info[2] == 0, "Cholesky must not fail"
    assert np.abs(p - rp).max() <= 1e-4 * max(1.0, np.abs(rp).max())
    assert np.abs(d - rd).max() <= 1e-4 * np.abs(rd).max()
    assert np.abs(k - rk).max() <= 1e-4 * np.abs(rk).max()
    assert np.abs(rp - g.poses).max() + np.abs(rd - g.disps).max() > 1e-3


def test_dense_ba_matches_reference_solver_at_the_headline_size():
    """BASELINE configs[2] without the oracle in between: the HIP BA on the very graph `bench.py` times (48 keyframes,
    48 x 64, E = 276, depth prior on, 3 Gauss-Newton iterations) against the reference `Solver`'s own output
    (`ba_headline_reference.npz`, make_golden.gen_headline), 1e-4 relative; the default kernel selection (fused
    matrix-core accumulate, two-chain band solve) and the general forms."""
    from vipe_amd.ext import slam_ext
    G = np.load(os.path.join(GOLD, "ba_headline_reference.npz"))
    g = make_graph(n=48, height=384, width=512, radius=3, seed=1234, depth_prior=True)
    assert len(g.ii) == int(G["n_edges"][0]) == 276
    rp, rd = G["poses"], G["disps"]
    for opts in (0, slam_ext.BA_OPT_ONE_CHAIN | slam_ext.BA_OPT_GENERAL_ACCUMULATE):
        p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole",
                                   dict(t0=1, t1=48, n_iters=3, pose_damping=1e-3, pose_ep=0.1, solver_options=opts))
        assert info[2] == 0, "Cholesky must not fail"
        assert np.abs(p - rp).max() <= 1e-4 * max(1.0, np.abs(rp).max()), opts
        assert np.abs(d - rd).max() <= 1e-4 * np.abs(rd).max(), opts
    assert np.abs(rp - g.poses).max() + np.abs(rd - g.disps).max() > 1e-3


def test_bench_operator_path_matches_reference_update_module_at_the_headline_grid():
    """SURVEY 8(c) golden #1 at [1,4,.,48,64] against the operator exactly as the bench's step drives it: natively
    sequenced (`vipe_update_operator`), gate context hoisted (`gate_context`), the hidden-state part of the z|r gates
    and the global-context terms staged beforehand (`hidden_gate_state`, for HALF the edges as `FactorGraph.update`
    does), two streams inside the call - vs the reference class's own fp32 outputs.  Tolerance as
    test_update_module_matches_reference_fixture (fp16 activations through ~8 layers)."""
    from vipe_amd.synth import headline_update_module_inputs
    from vipe_amd.slam.update_engine import CORR_CH, UpdateEngine, segment_csr
    G = np.load(os.path.join(GOLD, "update_module_headline_reference.npz"))
    um, net, inp, cor, flow = headline_update_module_inputs()
    eng = UpdateEngine(um, dev())
    eng.op_side_min_edges = 1  # the bench's E = 276 is above the default threshold; here 4 edges must take the same path
    E, ht, wd = 4, 48, 64
    f16 = torch.float16
    net_n = net[0].permute(0, 2, 3, 1).contiguous().to(dev(), f16)
    xbuf = torch.zeros(E, ht, wd, 320, dtype=f16, device=dev())
    xbuf[..., :128] = inp[0].permute(0, 2, 3, 1).to(dev(), f16)
    corr_n = torch.zeros(E, ht, wd, CORR_CH, dtype=f16, device=dev())
    corr_n[..., :196] = cor[0].permute(0, 2, 3, 1).to(dev(), f16)
    motn = flow[0].permute(0, 2, 3, 1).contiguous().to(dev(), f16)
    ix = torch.from_numpy(G["ix"]).to(dev())
    csr = segment_csr(ix, 3)
    pg = eng.gate_context(xbuf)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # staged on a second stream, as under the BA
        gs = eng.hidden_gate_state(net_n, pg, n_staged=2)
    torch.cuda.current_stream().wait_stream(side)
    assert eng.gate_state_matches(gs, net_n, pg)
    n2, dw, eta, _ = eng.forward_nhwc(net_n, xbuf, corr_n, motn, ix=ix, n_src=3, csr=csr, pgate=pg, gate_state=gs, native=True)
    torch.cuda.synchronize()
    assert eng._op_side is not None
    n2 = n2.permute(0, 3, 1, 2)[None].float().cpu().numpy()
    assert np.abs(n2[:, :, ::8] - G["out_net_sub"].astype(np.float32)).max() < 2e-2
    assert np.abs(dw[..., 0:2].cpu().numpy()[None] - G["out_delta"]).max() < 2e-2
    assert np.abs(dw[..., 2:4].cpu().numpy()[None] - G["out_weight"]).max() < 1e-2
    assert np.abs(eta.cpu().numpy()[None] - G["out_eta"]).max() < 2e-4


def _ba_rig_cases():
    src = open(os.path.join(GOLD, "make_golden.py")).read()
    ns = {}
    exec(src[src.index("BA_RIG_CASES = {"):src.index("def gen_ba_rig")], ns)
    return ns["BA_RIG_CASES"]


BA_RIG_CASES = _ba_rig_cases()


def run_hip_ba_rig(g, bk):
    """multi-view rig through the C ABI exactly as GraphBuffer.bundle_adjustment calls it (flattened (n v) disparities)"""
    from vipe_amd.ext import slam_ext
    from vipe_amd.synth import expand_edges
    pi, qi, di, pj, qj = expand_edges(g.ii, g.jj, g.V)
    M = len(pi)
    poses, intr, rig = T(g.poses).clone(), T(g.intrinsics).clone(), T(g.rig).clone()
    disps = T(g.disps.reshape(g.n * g.V, g.ht, g.wd)).clone()
    info = slam_ext.dense_ba(poses, disps, T(g.disps_sens.reshape(g.n * g.V, g.ht, g.wd)), intr, rig,
                             T(g.target.reshape(M, -1, 2)), T(g.weight.reshape(M, -1, 2)),
                             T(g.eta.reshape(g.n * g.V, g.ht, g.wd)), T(pi), T(qi), T(pj), T(qj), T(di), want_info=True, **bk)
    torch.cuda.synchronize()
    return (poses.cpu().numpy(), disps.cpu().numpy().reshape(g.n, g.V, g.ht, g.wd), intr.cpu().numpy(), rig.cpu().numpy(),
            info.cpu().numpy())


@pytest.mark.parametrize("name", sorted(BA_RIG_CASES))
def test_dense_ba_rig_matches_reference_solver(name):
    """Multi-view rigs in the fused BA (SURVEY 8 BA-diff "optional intrinsics / rig groups"): V = 2, 3, 5 and 6 views,
    cross-view self edges, one intrinsics block per view, the rig-rotation group (view 0 fixed, rotation-only
    retraction) - against the reference's own Solver (ba_rig_reference.npz) at the north_star tolerance, 1e-4 relative."""
    from vipe_amd.synth import make_rig_graph
    G = np.load(os.path.join(GOLD, "ba_rig_reference.npz"))
    gk, bk = BA_RIG_CASES[name]
    g = make_rig_graph(**gk)
    p, d, k, r, info = run_hip_ba_rig(g, dict(bk))
    rp, rd, rk, rr = (G[f"{name}/{x}"] for x in ("poses", "disps", "intrinsics", "rig"))
    assert info[2] == 0, "Cholesky must not fail"
    n_tail = (g.V * 1 if bk.get("optimize_intrinsics") else 0) + (6 * (g.V - 1) if bk.get("optimize_rig_rotation") else 0)
    assert info[3] == 6 * info[0] + n_tail
    assert np.abs(p - rp).max() <= 1e-4 * max(1.0, np.abs(rp).max()), np.abs(p - rp).max()
    assert np.abs(d - rd).max() <= 1e-4 * np.abs(rd).max(), np.abs(d - rd).max() / np.abs(rd).max()
    assert np.abs(k - rk).max() <= 1e-4 * np.abs(rk).max()
    assert np.abs(r - rr).max() <= 1e-4, np.abs(r - rr).max()
    assert np.array_equal(r[0], g.rig[0])
    if not bk.get("optimize_rig_rotation"):
        assert np.array_equal(r, g.rig)
    if not bk.get("optimize_intrinsics"):
        assert np.array_equal(k, g.intrinsics)
    assert np.abs(rp - g.poses).max() + np.abs(rd - g.disps).max() > 1e-3


def test_dense_ba_rig_at_bench_resolution_against_fp64_oracle():
    """A 2-camera rig at the BASELINE grid (48x64, 6 keyframes, radius 2 + cross-view terms, rig rotation + per-view
    intrinsics + sensor depth): fp32 HIP vs the fp64 oracle (itself pinned to the reference by the fixtures above)."""
    from vipe_amd.synth import make_rig_graph
    g = make_rig_graph(n=6, V=2, height=384, width=512, radius=2, seed=21, depth_prior=True)
    bk = dict(t0=1, t1=6, n_iters=2, pose_damping=1e-5, pose_ep=1e-2, optimize_intrinsics=True, optimize_rig_rotation=True)
    p, d, k, r, info = run_hip_ba_rig(g, bk)
    M = g.target.shape[0]
    op, od, ok_, orr = oba.bundle_adjustment(g.poses, g.disps, g.disps_sens, g.intrinsics, g.rig, g.target.reshape(M, -1, 2),
                                             g.weight.reshape(M, -1, 2), g.eta, g.ii, g.jj, **bk)
    assert info[2] == 0 and info[3] == 6 * 5 + 2 + 6
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(d - od).max() <= 1e-4 * np.abs(od).max()
    assert np.abs(k - ok_).max() <= 1e-4 * np.abs(ok_).max()
    assert np.abs(r - orr).max() <= 1e-4


def test_dense_ba_bench_size_against_fp64_oracle():
    """N=12, 48x64 grid (the BASELINE resolution), E=66: HIP fp32 vs the fp64 oracle."""
    g = make_graph(n=12, height=384, width=512, radius=3, seed=1234)
    bk = dict(t0=1, t1=12, n_iters=3, pose_damping=1e-3, pose_ep=0.1, motion_only=False, limited_disp=False,
              optimize_intrinsics=False)
    p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole", bk)
    E = len(g.ii)
    op, od, _, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                         ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                         g.eta[:, None], g.ii, g.jj, **bk)
    assert info[0] == 11 and info[3] == 66 and info[2] == 0
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()


def test_dense_ba_unused_buffer_rows_and_clamp():
    """Poses / frames beyond the graph are untouched except for the final disps.clamp_(min=1e-3) (buffer.py:525)."""
    from vipe_amd.ext import slam_ext
    g = make_graph(n=4, height=96, width=128, radius=2, seed=5)
    E = len(g.ii)
    NB = 9
    poses = torch.zeros(NB, 7, device=dev())
    poses[:, 6] = 1
    poses[:4] = T(g.poses)
    disps = torch.full((NB, g.ht, g.wd), 1e-5, device=dev())
    disps[:4] = T(g.disps)
    sens = torch.zeros_like(disps)
    eta = torch.full_like(disps, 1e-6)
    z = np.zeros_like(g.ii)
    before = poses.clone()
    slam_ext.dense_ba(poses, disps, sens, T(g.intrinsics), T(ose3.se3_identity(1)), T(g.target.reshape(E, -1, 2)),
                      T(g.weight.reshape(E, -1, 2)), eta, T(g.ii), T(z), T(g.jj), T(z), T(g.ii), 1, 4, 2, 1e-3, 0.1)
    assert torch.equal(poses[4:], before[4:]) and torch.equal(poses[0], before[0])
    assert not torch.equal(poses[1:4], before[1:4])
    assert float(disps[4:].min()) == pytest.approx(1e-3) and float(disps[4:].max()) == pytest.approx(1e-3)


def test_dense_ba_full_size_properties():
    """BASELINE size (N=48, 48x64, E=276): energy decreases monotonically over GN iterations and a second
    identical call from the same state reproduces the same result to fp32 reduction noise."""
    g = make_graph()  # 48 KFs, 512x384, E=276
    assert len(g.ii) == 276
    bk = dict(t0=1, t1=48, pose_damping=1e-3, pose_ep=0.1, motion_only=False, limited_disp=False,
              optimize_intrinsics=False)
    rig = ose3.se3_identity(1)
    e = [oba.energy(g.poses, g.disps[:, None], g.intrinsics, rig, g.target, g.weight, g.ii, g.jj)]
    for it in (1, 2, 3):
        p, d, _, info = run_hip_ba(g, g.intrinsics, "pinhole", dict(bk, n_iters=it))
        assert info[0] == 47 and info[3] == 282 and info[2] == 0
        e.append(oba.energy(p, d[:, None], g.intrinsics, rig, g.target, g.weight, g.ii, g.jj))
    assert e[1] < e[0] and e[2] < e[1] and e[3] < e[2]
    p2, d2, _, _ = run_hip_ba(g, g.intrinsics, "pinhole", dict(bk, n_iters=3))
    assert np.abs(p2 - p).max() < 1e-5 and np.abs(d2 - d).max() < 1e-4 * np.abs(d).max()


# ------------------------------------------------------------------------------------------------ flow-update operator


def _seeded_update_inputs(E, ht, wd):
    gen = torch.Generator().manual_seed(1)
    net = torch.randn(1, E, 128, ht, wd, generator=gen).tanh()
    inp = torch.randn(1, E, 128, ht, wd, generator=gen).relu()
    cor = torch.randn(1, E, 196, ht, wd, generator=gen)
    flow = torch.randn(1, E, 4, ht, wd, generator=gen) * 4
    return net, inp, cor, flow


def test_update_module_matches_reference_fixture():
    """MFMA convolutions (fp16 in, fp32 accumulate) vs the reference UpdateModule output (fp32 CPU fixture).
    Tolerance: fp16 activations through ~8 layers -> 2e-2 absolute on O(1) outputs (the reference itself runs
    these convolutions under fp16 autocast, factor_graph.py:230)."""
    from vipe_amd.slam.networks import UpdateModule
    from vipe_amd.slam.update_engine import UpdateEngine
    G = np.load(os.path.join(GOLD, "update_module_reference.npz"))
    torch.manual_seed(0)
    um = UpdateModule().eval()
    net, inp, cor, flow = _seeded_update_inputs(5, 12, 16)
    ix = torch.from_numpy(G["ix"])
    eng = UpdateEngine(um, dev())
    n2, delta, weight, eta, upmask = eng.forward(net.to(dev()).half(), inp.to(dev()).half(), cor.to(dev()).half(),
                                                 flow.to(dev()).half(), ix.to(dev()))
    torch.cuda.synchronize()
    assert np.abs(n2[:, :, ::4].float().cpu().numpy() - G["out_net_sub"]).max() < 2e-2
    assert np.abs(delta.float().cpu().numpy() - G["out_delta"]).max() < 2e-2
    assert np.abs(weight.float().cpu().numpy() - G["out_weight"]).max() < 1e-2
    assert np.abs(eta.float().cpu().numpy() - G["out_eta"]).max() < 2e-4
    assert np.abs(upmask[:, :, ::16].float().cpu().numpy() - G["out_upmask_sub"]).max() < 2e-2


def test_conv_mfma_against_torch_fp32():
    """Single convolutions through the C ABI vs torch fp32 conv2d of the same fp16-rounded operands."""
    import torch.nn.functional as F
    from vipe_amd._lib import check, lib, ptr, stream_ptr
    from vipe_amd.slam.update_engine import _Packed
    torch.manual_seed(3)
    for (B, H, W, cin, cout, k, act) in [(3, 12, 16, 128, 128, 3, "relu"), (2, 9, 7, 448, 128, 3, "none"),
                                         (2, 12, 16, 200, 128, 1, "relu"), (2, 10, 12, 4, 128, 7, "relu"),
                                         (1, 6, 8, 128, 64, 3, "tanh"), (4, 5, 5, 64, 576, 1, "sigmoid")]:
        x = (torch.randn(B, H, W, cin) * 0.5).half().to(dev())
        w = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).half()
        b = torch.randn(cout) * 0.1
        pk = _Packed(w, b, dev())
        y = torch.full((B, H, W, cout), 7.0, dtype=torch.float16, device=dev())
        check(lib().vipe_conv2d_nhwc_f16(ptr(x), ptr(pk.packed), ptr(pk.bias), None, ptr(y), B, H, W, cin, cin, 0, cout,
                                         cout, 0, k, k, {"none": 0, "relu": 1, "sigmoid": 2, "tanh": 3}[act],
                                         stream_ptr(x)), "conv")
        ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float(), b, padding=k // 2)
        ref = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](ref)
        err = (y.float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err < 4e-3, (cin, cout, k, err)


def test_conv_halo_tile_kernel_against_torch_fp32():
    """Width-64 images take the halo-tile kernel (4 rows x 64 px x 128 couts per workgroup): same check as above,
    including a two-source input (channel split) and ragged channel counts."""
    import torch.nn.functional as F
    from vipe_amd._lib import check, lib, ptr, stream_ptr
    from vipe_amd.slam.update_engine import _Packed
    torch.manual_seed(5)
    for (B, H, W, cin, cout, k, act) in [(2, 8, 64, 128, 128, 3, "relu"), (1, 4, 64, 448, 256, 3, "none"),
                                         (3, 12, 64, 200, 128, 1, "relu"), (2, 8, 64, 128, 384, 3, "tanh"),
                                         # config 5 (1024x512 -> 64x128 grid): two 64-column segments per row
                                         (1, 8, 128, 128, 128, 3, "relu"), (2, 4, 128, 200, 128, 1, "none"),
                                         (1, 8, 128, 4, 128, 7, "relu"), (2, 8, 128, 256, 4, 3, "none"),
                                         (1, 4, 192, 128, 64, 3, "sigmoid")]:
        x = (torch.randn(B, H, W, cin) * 0.5).half().to(dev())
        w = (torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).half()
        b = torch.randn(cout) * 0.1
        pk = _Packed(w, b, dev())
        y = torch.full((B, H, W, cout), 7.0, dtype=torch.float16, device=dev())
        check(lib().vipe_conv2d_nhwc_f16(ptr(x), ptr(pk.packed), ptr(pk.bias), None, ptr(y), B, H, W, cin, cin, 0, cout,
                                         cout, 0, k, k, {"none": 0, "relu": 1, "sigmoid": 2, "tanh": 3}[act],
                                         stream_ptr(x)), "conv")
        ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float(), b, padding=k // 2)
        ref = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](ref)
        err = (y.float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err < 4e-3, (cin, cout, k, err)
    # two-source input: channels [0,128) from one tensor, [128,448) from another (the GRU gate input)
    from vipe_amd.slam.update_engine import UpdateEngine  # noqa: F401
    B, H, W = 2, 8, 64
    xa = (torch.randn(B, H, W, 128) * 0.5).half().to(dev())
    xb = (torch.randn(B, H, W, 320) * 0.5).half().to(dev())
    w = (torch.randn(128, 448, 3, 3) / (448 * 9) ** 0.5).half()
    pk = _Packed(w, torch.zeros(128), dev())
    y = torch.empty((B, H, W, 128), dtype=torch.float16, device=dev())
    check(lib().vipe_conv2d_fused(ptr(xa), 128, 0, ptr(xb), 320, 0, 128, ptr(pk.packed), ptr(pk.bias), None, 0, 0, ptr(y),
                                  128, 0, None, 0, 0, None, 0, 0, None, None, None, 0, 0, B, H, W, 448, 128, 3, 3, 0, 0,
                                  stream_ptr(xa)),
          "conv_fused")
    ref = F.conv2d(torch.cat([xa, xb], -1).float().cpu().permute(0, 3, 1, 2), w.float(), None, padding=1)
    assert (y.float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 4e-3


# ------------------------------------------------------------------------------------------------ slam_ext geometry kernels


def test_frame_distance_depth_filter_projmap_iproj():
    from oracle import frame_ops
    from vipe_amd.ext import slam_ext
    g = make_graph(n=8, height=96, width=128, radius=2, seed=9)
    intr8 = (g.intrinsics / 8.0).astype(np.float32)
    ii = np.array([0, 1, 2, 5, 7, 3], dtype=np.int64)
    jj = np.array([1, 0, 6, 2, 4, 3], dtype=np.int64)
    z = np.zeros_like(ii)
    d = slam_ext.frame_distance(T(g.poses), T(g.disps), T(intr8), T(ii), T(jj), T(z), T(z), T(ii), 0.3)
    ref = frame_ops.frame_distance(g.poses, g.disps, intr8, ii, jj, z, z, ii, 0.3)
    assert np.abs(d.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    assert abs(float(d[5])) < 1e-5  # identical frames: zero induced flow
    # < 75 % valid -> 1000 (geom_kernels.cu:674): put the target camera far behind the source
    far = g.poses.copy()
    far[1, 2] = -100.0
    d2 = slam_ext.frame_distance(T(far), T(g.disps), T(intr8), T(ii[:1]), T(jj[:1]), T(z[:1]), T(z[:1]), T(ii[:1]), 0.3)
    assert float(d2[0]) == 1000.0
    # depth filter: integer counts, bit-exact
    inds = np.arange(8, dtype=np.int64)
    thresh = np.full(8, 0.05 / g.disps.mean(), dtype=np.float32)
    cnt = slam_ext.depth_filter(T(g.poses), T(g.disps), T(intr8[0]), T(inds), T(thresh))
    ref = frame_ops.depth_filter(g.poses, g.disps, intr8[0], inds, thresh)
    assert np.array_equal(cnt.cpu().numpy(), ref)  # the oracle follows the kernel's fp32 operation order (no contraction)
    assert cnt.max() <= 6
    coords, valid = slam_ext.projmap(T(g.poses), T(g.disps), T(intr8[0]), T(ii), T(jj))
    rc, rv = frame_ops.projmap(g.poses, g.disps, intr8[0], ii, jj)
    assert np.abs(coords.cpu().numpy() - rc).max() < 2e-3 and np.array_equal(valid.cpu().numpy(), rv)
    pts = slam_ext.iproj(T(g.poses), T(g.disps), T(intr8[0]))
    assert np.abs(pts.cpu().numpy() - frame_ops.iproj(g.poses, g.disps, intr8[0])).max() < 1e-3


def test_frame_distance_through_buffer_api():
    """GraphBuffer.frame_distance_dense_disp (buffer.py:550-593) against the independent oracle reprojection."""
    from vipe_amd.slam.buffer import GraphBuffer
    g = make_graph(n=6, height=96, width=128, radius=2, seed=10)
    buf = GraphBuffer(96, 128, buffer_size=8, device=dev())
    buf.n_frames = 6
    buf.poses[:6] = T(g.poses)
    buf.disps[:6, 0] = T(g.disps)
    buf.intrinsics[:] = T(g.intrinsics)
    ii, jj = T(np.array([0, 2, 5])), T(np.array([1, 4, 3]))
    d = buf.frame_distance_dense_disp(ii, jj, beta=0.3, bidirectional=False).cpu().numpy()[:, 0]
    z = np.zeros(3, np.int64)
    o = ogeom.reproject(g.poses, g.disps, (g.intrinsics / 8).astype(np.float32), ose3.se3_identity(1),
                        np.array([0, 2, 5]), np.array([1, 4, 3]), z, z, np.array([0, 2, 5]))
    u, v = ogeom.pixel_grid(g.ht, g.wd, np.float32)
    full = np.sqrt((o["coords"][..., 0] - u) ** 2 + (o["coords"][..., 1] - v) ** 2).mean((1, 2))
    assert np.all(d > 0.3 * full * 0.99)  # beta-weighted full-motion part is a lower bound


# ------------------------------------------------------------------------------------------------ altcorr / scatter / corr_ext


def test_altcorr_forward_matches_oracle_and_volume_lookup():
    from vipe_amd.ext import droid_net_ext
    rng = np.random.default_rng(11)
    B, H, W, C = 3, 8, 16, 128
    f1 = rng.normal(0, 1, (B, H, W, C)).astype(np.float32)
    f2 = rng.normal(0, 1, (B, H // 2, W // 2, C)).astype(np.float32)
    coords = np.stack([rng.uniform(-2, W // 2 + 1, (B, 2, H, W)), rng.uniform(-2, H // 2 + 1, (B, 2, H, W))], -1).astype(np.float32)
    (out,) = droid_net_ext.altcorr_forward(T(f1), T(f2), T(coords), 3)
    ref = ocorr.altcorr_forward(f1, f2, coords, 3)
    assert out.shape == (B, 2, 49, H, W)
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    # the tiled kernel's paths: smooth flow on a ragged grid (windows of a 4 x 8 tile staged in LDS as one box), tiles
    # whose windows are scattered over the whole map (direct taps), tiles entirely outside it, channel counts that are not
    # multiples of 32 / 4, and fp16 maps (fp32 accumulation)
    for (H1, W1, H2, W2, C, spread) in ((41, 73, 41, 73, 128, 1.5), (10, 18, 5, 9, 128, 1.0), (13, 11, 40, 60, 96, 30.0),
                                        (6, 9, 12, 12, 50, 2.0)):
        f1 = rng.normal(0, 1, (2, H1, W1, C)).astype(np.float32)
        f2 = rng.normal(0, 1, (2, H2, W2, C)).astype(np.float32)
        u, v = np.meshgrid(np.arange(W1, dtype=np.float32) * (W2 / W1), np.arange(H1, dtype=np.float32) * (H2 / H1))
        coords = (np.stack([u, v], -1)[None, None] + rng.normal(0, spread, (2, 2, H1, W1, 2))).astype(np.float32)
        coords[1, 1] += 500.0  # one coordinate set far outside the map: all zeros
        (out,) = droid_net_ext.altcorr_forward(T(f1), T(f2), T(coords), 3)
        ref = ocorr.altcorr_forward(f1, f2, coords, 3)
        assert np.abs(out.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max(), (H1, W1, C)
        assert not out[1, 1].any()
        (outh,) = droid_net_ext.altcorr_forward(T(f1).half(), T(f2).half(), T(coords), 3)
        refh = ocorr.altcorr_forward(f1.astype(np.float16).astype(np.float32), f2.astype(np.float16).astype(np.float32), coords, 3)
        assert np.abs(outh.float().cpu().numpy() - refh).max() <= 2e-3 * np.abs(refh).max(), (H1, W1, C)


def test_altcorr_block_matches_corr_block():
    """AltCorrBlock (volume-free) == CorrBlock (volume) on the same features, up to fp16 volume rounding."""
    from vipe_amd.slam.networks import AltCorrBlock, CorrBlock
    rng = np.random.default_rng(12)
    N, h, w = 3, 16, 16
    fm = torch.from_numpy(rng.normal(0, 1, (1, N, 128, h, w)).astype(np.float16)).to(dev())
    ii, jj = torch.tensor([0, 1, 2], device=dev()), torch.tensor([1, 2, 0], device=dev())
    coords = torch.from_numpy(np.stack([rng.uniform(0, w - 1, (1, 3, h, w)), rng.uniform(0, h - 1, (1, 3, h, w))], -1)
                              .astype(np.float32)).to(dev())
    a = AltCorrBlock(fm)(coords, ii, jj)
    c = CorrBlock(fm[:, ii], fm[:, jj])(coords)
    assert a.shape == c.shape == (1, 3, 196, h, w)
    assert (a.float() - c.float()).abs().max().item() < 0.05


def test_scatter_ext_against_torch():
    """All five reductions reach the HIP kernel (`vipe_scatter`) - sum and mean included - and their autograd adjoints
    (scatter.cpp:38-201) match torch.scatter_reduce's."""
    from vipe_amd.ext import scatter, scatter_ext
    torch.manual_seed(0)
    for dtype, tol in ((torch.float32, 1e-5), (torch.float64, 1e-12)):
        src = torch.randn(4, 33, 6, device=dev(), dtype=dtype, requires_grad=True)
        idx = torch.randint(0, 9, (33,), device=dev())
        full = idx.view(1, -1, 1).expand(4, 33, 6)
        used = torch.zeros(10, dtype=torch.bool, device=dev())
        used[idx] = True
        for red, tr, init in (("sum", "sum", 0.0), ("mean", "mean", 0.0), ("mul", "prod", 1.0),
                              ("min", "amin", float("inf")), ("max", "amax", float("-inf"))):
            out = scatter.scatter(src, idx, dim=1, dim_size=10, reduce=red)
            ref = torch.full((4, 10, 6), init, device=dev(), dtype=dtype).scatter_reduce(
                1, full, src, tr, include_self=tr in ("sum", "prod"))
            assert out.shape == (4, 10, 6)
            assert torch.allclose(out[:, used], ref[:, used], rtol=tol * 10, atol=tol), red
            assert (out[:, ~used] == (1.0 if red == "mul" else 0.0)).all(), red  # untouched slots
            go = torch.randn_like(out)
            (g1,) = torch.autograd.grad(out, src, go, retain_graph=True)
            (g2,) = torch.autograd.grad(ref, src, go)
            assert torch.allclose(g1, g2, rtol=tol * 100, atol=tol * 10), red
        mx, arg = scatter.scatter_max(src, idx, dim=1, dim_size=10)
        assert arg.dtype == torch.int64 and (arg[:, ~used] == 33).all()
        assert torch.equal(src.detach().gather(1, arg[:, used].clamp(max=32)), mx[:, used].detach())
    # fp16 rows (what the flow-update operator would hand over), 1-D and full-shape indices, negative dim, `out=`
    h = torch.randn(7, 16, device=dev()).half()
    ix = torch.tensor([0, 0, 1, 2, 2, 2, 4], device=dev())
    sm = scatter.scatter_mean(h, ix, dim=0, dim_size=5)
    ref = torch.zeros(5, 16, device=dev()).index_add_(0, ix, h.float()) / torch.tensor([2, 1, 3, 1, 1], device=dev())[:, None]
    assert (sm.float() - ref).abs().max() < 4e-3
    acc = torch.ones(5, 16, device=dev())
    r = scatter_ext.scatter_sum(h.float(), ix.view(-1, 1).expand(7, 16), -2, acc, None)
    assert r.data_ptr() == acc.data_ptr() and torch.allclose(acc, 1 + torch.zeros(5, 16, device=dev()).index_add_(0, ix, h.float()))
    with pytest.raises(RuntimeError):
        scatter.scatter_sum(h.float(), ix.int(), 0)  # index must be int64
    # integer sources (Tensor.scatter_add_ in the reference's Python layer): torch's scatter, same values
    hi = torch.randint(-5, 6, (7, 16), device=dev())
    assert torch.equal(scatter.scatter_sum(hi, ix, dim=0, dim_size=5), torch.zeros(5, 16, dtype=hi.dtype, device=dev()).index_add_(0, ix, hi))
    # a slot index beyond `out` writes nothing out of bounds: the rows in range land, the canary behind `out` survives
    big = torch.zeros(6, 16, device=dev())
    big[5] = 9.0
    scatter_ext.scatter_sum(h.float(), torch.tensor([0, 7, 1, 2, 9, 2, 4], device=dev()), 0, big[:5], None)
    assert (big[5] == 9.0).all() and torch.allclose(big[0], h[0].float()) and torch.allclose(big[2], (h[3] + h[5]).float(), atol=1e-3)
    # the mean's row count comes from the [E] index, not from an int64 copy of src's shape: wide rows stay cheap
    wide = torch.randn(7, 3, 5, 4, device=dev())
    sm = scatter.scatter_mean(wide, ix, dim=0, dim_size=5)
    refw = torch.zeros(5, 3, 5, 4, device=dev()).index_add_(0, ix, wide) / torch.tensor([2, 1, 3, 1, 1.0], device=dev()).view(5, 1, 1, 1)
    assert torch.allclose(sm, refw, atol=1e-6)


def test_corr_ext_sampler_against_torch_loops():
    from vipe_amd.ext import corr_ext
    torch.manual_seed(14)
    B, C, H, W = 2, 5, 7, 9
    a = torch.randn(B, C, H, W, device=dev())
    b = torch.randn(B, C, H, W, device=dev())
    kH = kW = 1
    patch = 5
    out = corr_ext.forward(a, b, kH, kW, patch, patch, 0, 0, 1, 1, 1, 1, 1, 1)
    assert out.shape == (B, patch, patch, H, W)
    bp = torch.nn.functional.pad(b, (2, 2, 2, 2))
    ref = torch.stack([torch.stack([(a * bp[:, :, ph:ph + H, pw:pw + W]).sum(1) for pw in range(patch)], 1) for ph in range(patch)], 1)
    assert torch.allclose(out, ref, atol=1e-5)
    g = torch.randn_like(out)
    g1, g2 = corr_ext.backward(a, b, g, kH, kW, patch, patch, 0, 0, 1, 1, 1, 1, 1, 1)
    a_ = a.clone().requires_grad_(True)
    b_ = b.clone().requires_grad_(True)
    bp = torch.nn.functional.pad(b_, (2, 2, 2, 2))
    ref = torch.stack([torch.stack([(a_ * bp[:, :, ph:ph + H, pw:pw + W]).sum(1) for pw in range(patch)], 1) for ph in range(patch)], 1)
    (ref * g).sum().backward()
    assert torch.allclose(g1, a_.grad, atol=1e-4) and torch.allclose(g2, b_.grad, atol=1e-4)


def test_corr_ext_device_kernels_match_the_reference_cpu_implementation():
    """The HIP sampler kernels vs outputs of the reference's own CPU implementation (csrc/corr_ext/correlation.cpp compiled
    in the build container, tests/golden/corr_sampler_reference.npz): forward and backward on every geometry with an odd
    patch size; for the EVEN patch the reference's CUDA forward kernel centres the patch differently from its CPU code
    and from its own CUDA backward kernels (correlation_cuda_kernel.cu:43-51 vs :96-117 / correlation.cpp:73-74) - the
    device forward follows the CUDA forward (checked against that formula), the device backward the fixture."""
    from test_abi import _corr_sampler_cases
    from vipe_amd.ext import corr_ext
    G = np.load(os.path.join(GOLD, "corr_sampler_reference.npz"))
    for name, (shape, geom) in _corr_sampler_cases().items():
        a, b, go = (T(G[f"{name}/{k}"]) for k in ("a", "b", "grad_out"))
        out = corr_ext.forward(a, b, *geom)
        if name != "even_patch":
            assert np.allclose(out.cpu().numpy(), G[f"{name}/out"], atol=1e-4), name
        else:
            kH, kW, pH, pW, padH, padW, dilH, dilW, dpH, dpW, dH, dW = geom
            assert (kH, kW, padH, padW, dH, dW) == (1, 1, 0, 0, 1, 1)
            B, C, H, W = shape
            want = torch.zeros(B, pH, pW, H, W)
            ac, bc = a.cpu(), b.cpu()
            for ph in range(pH):
                for pw in range(pW):
                    sy, sx = ph * dpH - dpH * (pH - 1) // 2, pw * dpW - dpW * (pW - 1) // 2
                    for y in range(H):
                        for x in range(W):
                            if 0 <= y + sy < H and 0 <= x + sx < W:
                                want[:, ph, pw, y, x] = (ac[:, :, y, x] * bc[:, :, y + sy, x + sx]).sum(1)
            assert torch.allclose(out.cpu(), want, atol=1e-4)
        g1, g2 = corr_ext.backward(a, b, go, *geom)
        assert np.allclose(g1.cpu().numpy(), G[f"{name}/grad1"], atol=1e-4), name
        assert np.allclose(g2.cpu().numpy(), G[f"{name}/grad2"], atol=1e-4), name


def test_corr_pyramid_lookup_row_kernel_bit_exact():
    """Widths that are multiples of 64 take the 8-lanes-per-pixel kernel: bit-exact vs the oracle, both layouts,
    windows hanging over every border."""
    from vipe_amd.ext import droid_net_ext
    rng = np.random.default_rng(21)
    E, h, w = 2, 8, 64
    levels = [T(rng.normal(0, 1, (E, h, w, h >> i, w >> i)).astype(np.float16)) for i in range(4)]
    coords = np.stack([rng.uniform(-6, w + 5, (E, h, w)), rng.uniform(-6, h + 5, (E, h, w))], -1).astype(np.float32)
    coords[0, 0, :8] = np.stack([np.arange(8) * 8.0, np.arange(8) * 1.0], -1)  # integer coords, every chunk phase
    ref = ocorr.corr_lookup([lv.cpu().numpy() for lv in levels], coords[None], 3)[0]  # [E,196,h,w]
    out = droid_net_ext.corr_pyramid_lookup(levels, T(coords), 3)
    assert np.array_equal(out.cpu().numpy().view(np.uint16), ref.view(np.uint16))
    nhwc = droid_net_ext.corr_pyramid_lookup_nhwc(levels, T(coords), 3, 200)
    assert torch.equal(nhwc[..., :196].permute(0, 3, 1, 2), out) and torch.count_nonzero(nhwc[..., 196:]) == 0


# ------------------------------------------------------------------------------------------------ factor graph (composition)


def _tiny_graph(n=5, seed=41):
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.factor_graph import FactorGraph
    from vipe_amd.slam.networks import UpdateModule
    g = make_graph(n=n, height=128, width=128, radius=2, seed=seed)  # 16x16 grid: 4 pyramid levels exist
    buf = GraphBuffer(128, 128, buffer_size=8, device=dev())
    buf.n_frames = n
    buf.poses[:n] = T(g.poses)
    buf.disps[:n, 0] = T(g.disps)
    buf.intrinsics[:] = T(g.intrinsics)
    gen = torch.Generator().manual_seed(seed)
    buf.fmaps[:n, 0] = torch.randn(n, 128, 16, 16, generator=gen).half().to(dev())
    buf.nets[:n, 0] = torch.randn(n, 128, 16, 16, generator=gen).tanh().half().to(dev())
    buf.inps[:n, 0] = torch.randn(n, 128, 16, 16, generator=gen).relu().half().to(dev())
    torch.manual_seed(0)
    um = UpdateModule().eval()
    graph = FactorGraph(um, buf, dev(), max_factors=-1)
    graph.add_factors(torch.from_numpy(g.ii), torch.from_numpy(g.jj))
    return g, buf, graph, um


def test_factor_graph_update_matches_oracle_composition():
    """One FactorGraph.update (factor_graph.py:231-314) against the same iteration composed from oracle pieces:
    reproject -> pyramid lookup (bit-exact input to the GRU) -> fp32 UpdateModule -> fp64 BA.  The GRU runs in fp16
    on the device, so targets/weights agree to fp16 noise and the BA result to a few 1e-3 of the step."""
    from oracle import update_module as oum
    g, buf, graph, um = _tiny_graph()
    n, E = 5, len(g.ii)
    poses0, disps0 = buf.poses[:n].cpu().numpy().copy(), buf.disps[:n, 0].cpu().numpy().copy()
    target0 = graph.target[0].cpu().numpy().copy()
    levels = [lv.cpu().numpy() for lv in graph.corr.corr_pyramid]
    net0 = graph.f_net.float().cpu()
    inp0 = graph.inp.float().cpu()
    graph.update(t0=1, t1=n, itrs=2)
    torch.cuda.synchronize()
    # ---- oracle composition
    z = np.zeros_like(g.ii)
    intr8 = (g.intrinsics / 8.0).astype(np.float32)
    rig = ose3.se3_identity(1)
    o = ogeom.reproject(poses0, disps0, intr8, rig, g.ii, g.jj, z, z, g.ii)
    coords1 = o["coords"]
    u, v = ogeom.pixel_grid(16, 16, np.float32)
    grid = np.stack([u, v], -1)
    motn = np.clip(np.concatenate([coords1 - grid, target0 - coords1], -1).transpose(0, 3, 1, 2), -64, 64)
    corr = ocorr.corr_lookup(levels, coords1[None], 3)  # fp16, bit-exact with the device lookup
    ix = torch.from_numpy(np.unique(g.ii, return_inverse=True)[1])
    sd = {k: w.float() for k, w in um.state_dict().items()}
    with torch.no_grad():
        net2, delta, weight, eta, _ = oum.update_forward(sd, net0, inp0, torch.from_numpy(corr).float(),
                                                         torch.from_numpy(motn).float()[None].half().float(), ix)
    tgt = coords1 + delta[0].numpy()
    assert np.abs(graph.target[0].cpu().numpy() - tgt).max() < 0.05
    assert np.abs(graph.weight[0].cpu().numpy() - weight[0].numpy()).max() < 0.02
    assert np.abs(graph.f_net.float().cpu().numpy() - net2.numpy()).max() < 0.03
    damping = np.full((n, 1, 16, 16), 1e-6)
    damping[np.unique(g.ii), 0] = eta[0].numpy()
    op, od, _, _ = oba.bundle_adjustment(poses0, disps0[:, None], np.zeros_like(disps0)[:, None], g.intrinsics, rig,
                                         graph.target[0].cpu().numpy().reshape(E, -1, 2),
                                         graph.weight[0].cpu().numpy().reshape(E, -1, 2), damping, g.ii, g.jj, t0=1, t1=n,
                                         n_iters=2, pose_damping=1e-3, pose_ep=0.1)
    # same targets/weights (device values) into the oracle BA -> isolates the BA: 1e-4 relative
    assert np.abs(buf.poses[:n].cpu().numpy() - op).max() < 2e-4
    assert np.abs(buf.disps[:n, 0].cpu().numpy() - od[:, 0]).max() < 2e-4 * np.abs(od).max() + 1e-3 * np.abs(
        graph.damping[:n].cpu().numpy() - damping[:, 0]).max()
    assert int(graph.age.min()) == 1


def test_factor_graph_update_full_size_with_sensor_depth():
    """BASELINE configs[2] as bench.py runs it: ONE FactorGraph.update at N = 48, E = 276, 48x64, sensor-depth prior on
    every keyframe ("depth_align on").  (i) the lookup on the DEVICE-built pyramid (through the CorrPool slot
    indirection) is bit-exact vs the oracle for a sample of edges; (ii) the BA step equals the fp64 oracle BA fed the
    device's own targets / weights / damping at 1e-4 relative (north_star tolerance); (iii) the energy of those factors
    decreases; (iv) the sensor prior is live (the result differs from a run without it)."""
    import bench
    from vipe_amd.ext import slam_ext
    n = 48
    g, buf, graph = bench.build_problem(dev(), n, 384, 512, 3, 0, seed=1234, depth_prior=True)
    E = len(g.ii)
    assert E == 276 and float(buf.disps_sens[:n].sum()) > 0
    poses0, disps0 = buf.poses[:n].cpu().numpy().copy(), buf.disps[:n, 0].cpu().numpy().copy()
    # (i) lookup, sample of 12 edges spread over the graph
    z = torch.zeros(E, dtype=torch.long, device=dev())
    coords1, _ = slam_ext.reproject(buf.poses, buf.flattened_disps, buf.intrinsics, buf.rig, graph.ii, z, graph.jj, z,
                                    graph.ii)
    sel = list(range(0, E, 23))
    out = graph.corr.lookup_nhwc(coords1)  # [E,h,w,200] through the pool's slot vector
    lv_all = graph.corr.corr_pyramid
    levels = [lv[sel].cpu().numpy() for lv in lv_all]
    ref = ocorr.corr_lookup(levels, coords1[sel].cpu().numpy()[None], 3)[0]  # [12,196,h,w]
    got = out[sel][..., :196].permute(0, 3, 1, 2).cpu().numpy()
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), "full-size lookup on the device pyramid not bit-exact"
    del out, lv_all
    # the update itself
    graph.update(t0=1, t1=n, itrs=3)
    torch.cuda.synchronize()
    tg, wg = graph.target[0].cpu().numpy(), graph.weight[0].cpu().numpy()
    damping = graph.damping[:n].cpu().numpy()
    p1, d1 = buf.poses[:n].cpu().numpy(), buf.disps[:n, 0].cpu().numpy()
    assert np.isfinite(p1).all() and np.isfinite(d1).all() and np.isfinite(tg).all()
    # (ii) fp64 oracle BA on the device's targets / weights / eta, from the pre-update state
    rig = ose3.se3_identity(1)
    kw = dict(t0=1, t1=n, n_iters=3, pose_damping=1e-3, pose_ep=0.1)
    op, od, _, _ = oba.bundle_adjustment(poses0, disps0[:, None], g.disps_sens[:, None], g.intrinsics, rig,
                                         tg.reshape(E, -1, 2), wg.reshape(E, -1, 2), damping[:, None], g.ii, g.jj, **kw)
    assert np.abs(p1 - op).max() <= 1e-4 * max(1.0, np.abs(op).max()), np.abs(p1 - op).max()
    assert np.abs(d1 - od[:, 0]).max() <= 1e-4 * np.abs(od).max(), np.abs(d1 - od[:, 0]).max()
    # (iii) energy of the factors the BA was given
    e0 = oba.energy(poses0, disps0[:, None], g.intrinsics, rig, tg, wg, g.ii, g.jj)
    e1 = oba.energy(p1, d1[:, None], g.intrinsics, rig, tg, wg, g.ii, g.jj)
    assert e1 < e0
    # (iv) same BA without the prior moves the disparities elsewhere
    pz, dz = T(poses0).clone(), T(disps0).clone()
    zz = np.zeros_like(g.ii)
    slam_ext.dense_ba(pz, dz, torch.zeros_like(dz), T(g.intrinsics), T(rig), T(tg.reshape(E, -1, 2)),
                      T(wg.reshape(E, -1, 2)), T(damping), T(g.ii), T(zz), T(g.jj), T(zz), T(g.ii), 1, n, 3, 1e-3, 0.1)
    assert np.abs(dz.cpu().numpy() - d1).max() > 10 * 1e-4 * np.abs(od).max()


def test_factor_graph_on_a_two_camera_rig():
    """V = 2 through the host classes (GraphBuffer n_views = 2, FactorGraph cross_view): edge expansion incl. the
    cross-view self edges (buffer.py:318-361), per-(frame, view) hidden state / correlation / damping indexing, one
    `update` (frontend flags: rig and intrinsics fixed) whose BA equals the fp64 oracle fed the device's own targets /
    weights / damping, then `update_batch` with the rig-rotation and intrinsics groups switched on (backend flags)."""
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.factor_graph import FactorGraph
    from vipe_amd.slam.networks import UpdateModule
    from vipe_amd.synth import expand_edges, make_rig_graph
    g = make_rig_graph(n=4, V=2, height=128, width=128, radius=2, seed=33)
    n, V = g.n, g.V
    buf = GraphBuffer(128, 128, n_views=V, buffer_size=6, device=dev())
    buf.n_frames = n
    buf.poses[:n] = T(g.poses)
    buf.disps[:n] = T(g.disps)
    buf.intrinsics[:] = T(g.intrinsics)
    buf.rig[:] = T(g.rig)
    gen = torch.Generator().manual_seed(5)
    buf.fmaps[:n] = torch.randn(n, V, 128, 16, 16, generator=gen).half().to(dev())
    buf.nets[:n] = torch.randn(n, V, 128, 16, 16, generator=gen).tanh().half().to(dev())
    buf.inps[:n] = torch.randn(n, V, 128, 16, 16, generator=gen).relu().half().to(dev())
    torch.manual_seed(0)
    graph = FactorGraph(UpdateModule().eval(), buf, dev(), max_factors=-1, cross_view=True)
    graph.add_factors(torch.from_numpy(g.ii), torch.from_numpy(g.jj))
    pi, qi, di, pj, qj = expand_edges(g.ii, g.jj, V)
    M = len(pi)
    P_ = graph._edge_plan()
    for name, want in (("pi", pi), ("qi", qi), ("di", di), ("pj", pj), ("qj", qj)):
        assert np.array_equal(P_[name].cpu().numpy(), want), name
    assert graph.net_n.shape[0] == M and graph.target.shape[1] == M and len(graph.corr) == M
    # hidden state of term (edge e, view v) = nets[pi, qi]
    assert torch.equal(graph.net_n[5].permute(2, 0, 1), buf.nets[int(pi[5]), int(qi[5])])
    graph.target = T(g.target)[None].contiguous()
    graph.weight = T(g.weight)[None].contiguous()
    poses0, disps0 = buf.poses[:n].cpu().numpy().copy(), buf.disps[:n].cpu().numpy().copy()
    graph.update(t0=1, t1=n, itrs=2)
    torch.cuda.synchronize()
    tg, wg = graph.target[0].cpu().numpy(), graph.weight[0].cpu().numpy()
    damping = graph.damping.view(-1, V, 16, 16)[:n].cpu().numpy()
    kw = dict(t0=1, t1=n, n_iters=2, pose_damping=1e-3, pose_ep=0.1)
    op, od, ok_, orr = oba.bundle_adjustment(poses0, disps0, g.disps_sens, g.intrinsics, g.rig, tg.reshape(M, -1, 2),
                                             wg.reshape(M, -1, 2), damping, g.ii, g.jj, **kw)
    assert np.abs(buf.poses[:n].cpu().numpy() - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(buf.disps[:n].cpu().numpy() - od).max() <= 1e-4 * np.abs(od).max()
    assert np.array_equal(buf.rig.cpu().numpy(), g.rig) and np.array_equal(buf.intrinsics.cpu().numpy(), g.intrinsics)
    # backend pass: rig rotation + per-view intrinsics
    p1, d1 = buf.poses[:n].cpu().numpy().copy(), buf.disps[:n].cpu().numpy().copy()
    graph.update_batch(itrs=3, steps=1, optimize_intrinsics=True, optimize_rig_rotation=True)
    torch.cuda.synchronize()
    tg, wg = graph.target[0].cpu().numpy(), graph.weight[0].cpu().numpy()
    damping = graph.damping.view(-1, V, 16, 16)[:n].cpu().numpy()
    kw = dict(t0=1, t1=n, n_iters=3, pose_damping=1e-5, pose_ep=1e-2, optimize_intrinsics=True, optimize_rig_rotation=True)
    op, od, ok_, orr = oba.bundle_adjustment(p1, d1, g.disps_sens, g.intrinsics, g.rig, tg.reshape(M, -1, 2),
                                             wg.reshape(M, -1, 2), damping, g.ii, g.jj, **kw)
    assert np.abs(buf.poses[:n].cpu().numpy() - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(buf.disps[:n].cpu().numpy() - od).max() <= 1e-4 * np.abs(od).max()
    assert np.abs(buf.rig.cpu().numpy() - orr).max() <= 1e-4 and np.abs(buf.rig[1, 3:].cpu().numpy() - g.rig[1, 3:]).max() > 1e-6
    assert np.abs(buf.intrinsics.cpu().numpy() - ok_).max() <= 1e-4 * np.abs(ok_).max()
    e0 = oba.energy(p1, d1, g.intrinsics, g.rig, tg, wg, g.ii, g.jj)
    e1 = oba.energy(buf.poses[:n].cpu().numpy(), buf.disps[:n].cpu().numpy(), buf.intrinsics.cpu().numpy(),
                    buf.rig.cpu().numpy(), tg, wg, g.ii, g.jj)
    assert e1 < e0


def test_factor_graph_update_batch_runs_and_reduces_energy():
    g, buf, graph, um = _tiny_graph(seed=43)
    n = 5
    graph.target = T(g.target)[None].contiguous()
    graph.weight = T(g.weight)[None].contiguous()
    rig = ose3.se3_identity(1)
    graph.update_batch(itrs=4, steps=1, optimize_intrinsics=False, optimize_rig_rotation=False)
    torch.cuda.synchronize()
    p, d = buf.poses[:n].cpu().numpy(), buf.disps[:n].cpu().numpy()
    tg, wg = graph.target[0].cpu().numpy(), graph.weight[0].cpu().numpy()
    assert np.isfinite(p).all() and np.isfinite(d).all()
    # the BA step decreased the energy of the factors it was given
    e_before = oba.energy(g.poses, g.disps[:, None], g.intrinsics, rig, tg, wg, g.ii, g.jj)
    e_after = oba.energy(p, d, g.intrinsics, rig, tg, wg, g.ii, g.jj)
    assert e_after < e_before


def test_dense_ba_dense_window_takes_lds_dense_solver():
    """The frontend's windows: ~20 free poses, every pair coupled (proximity + inactive edges) - too wide for the LDS
    band solver, but the packed lower triangle of the reduced system (n = 114, and 115 with the focal length) fits LDS:
    `ba_solve_dense_kernel` (info[5] == 2) must give the fp64 oracle's answer."""
    g = make_graph(n=20, height=96, width=128, radius=19, seed=91)  # all ordered pairs: E = 380
    assert len(g.ii) == 380
    for intr in (False, True):
        bk = dict(t0=1, t1=20, n_iters=2, pose_damping=1e-3, pose_ep=0.1, motion_only=False, limited_disp=False,
                  optimize_intrinsics=intr)
        p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole", bk)
        E = len(g.ii)
        op, od, ok_, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                               ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                               g.eta[:, None], g.ii, g.jj, **bk)
        assert info[0] == 19 and info[3] == 114 + int(intr) and info[2] == 0 and info[4] == 18
        assert info[5] == 2, "the LDS dense solver should have taken this system"
        assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
        assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()
        assert np.abs(k - ok_).max() <= 1e-4 * np.abs(ok_).max()


@pytest.mark.parametrize("kind", ["band", "dense", "global"])
def test_dense_ba_path_hints_across_calls_with_one_plan(kind):
    """The caller-side plan state of `slam_ext.dense_ba` (`state=`, `plan_key=`): call 1 launches every accumulate / solve
    kernel and learns from d_info[4..7] which ones the plan uses; later calls with the SAME key reuse the plan and pass
    the hint, so the kernels that would only start to exit are not launched at all.  For a band-solved chain, a dense
    window (band solver declines, LDS dense solver takes it) and a system too large for LDS (both decline, tiled
    Cholesky): every call - before the hint exists, while it is pending, with it - starts from the same state and must
    give the fp64 oracle's answer with the same solver (info[5]: 1 band, 2 dense, 0 global)."""
    import time
    from vipe_amd.ext import slam_ext
    g = {"band": lambda: make_graph(n=14, height=96, width=128, radius=2, seed=71),
         "dense": lambda: make_graph(n=20, height=96, width=128, radius=19, seed=91),
         "global": lambda: make_graph(n=70, height=96, width=128, radius=3, extra_edges=260, seed=101)}[kind]()
    n, E = len(g.poses), len(g.ii)
    bk = dict(t0=1, t1=n, n_iters=2, pose_damping=1e-3, pose_ep=0.1)
    op, od, _, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics, ose3.se3_identity(1),
                                         g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2), g.eta[:, None], g.ii, g.jj, **bk)
    z = np.zeros_like(g.ii)
    idx = [T(x) for x in (g.ii, z, g.jj, z, g.ii)]  # the SAME index tensors every call (their addresses are part of the key)
    tgt, wgt, eta, sens = T(g.target.reshape(E, -1, 2)), T(g.weight.reshape(E, -1, 2)), T(g.eta), T(g.disps_sens)
    intr, rig = T(g.intrinsics), T(ose3.se3_identity(1))
    state, want_solver, hints_seen = {}, {"band": 1, "dense": 2, "global": 0}[kind], []
    for call in range(4):
        poses, disps = T(g.poses).clone(), T(g.disps).clone()
        info = slam_ext.dense_ba(poses, disps, sens, intr, rig, tgt, wgt, eta, *idx, want_info=True, state=state,
                                 plan_key=("one plan",), **bk)
        torch.cuda.synchronize()
        time.sleep(0.01)  # let the learnt facts arrive (pinned buffer + event)
        info = info.cpu().numpy()
        hints_seen.append(slam_ext._path_hint(state, state.get("key")))
        assert info[2] == 0 and info[5] == want_solver, (kind, call, info)
        assert np.abs(poses.cpu().numpy() - op).max() <= 1e-4 * max(1.0, np.abs(op).max()), (kind, call)
        assert np.abs(disps.cpu().numpy() - od[:, 0]).max() <= 1e-4 * np.abs(od).max(), (kind, call)
    assert hints_seen[-1] != 0, "the plan's kernel selection must have been learnt by the last call"
    solved_bits = {1: 4 | 16, 2: 4 | 8, 0: 8 | 16}[want_solver]
    assert hints_seen[-1] & solved_bits == solved_bits


@pytest.mark.parametrize("intr", [False, True])
def test_dense_ba_large_dense_system_takes_tiled_cholesky(intr):
    """The global BA's reduced systems (hundreds of coupled poses): 64 x 64 tiled Cholesky spread over the chip (potrf in
    one wave's registers + tile inverse, trsm / syrk on the fp64 matrix cores, rhs row riding along, tiled back
    substitution).  n = 414 (+1 with the focal length: a one-column last tile... 415 = 6 x 64 + 31), loop-closure-like
    long-range edges, against the fp64 oracle."""
    g = make_graph(n=70, height=96, width=128, radius=3, extra_edges=260, seed=101)
    bk = dict(t0=1, t1=70, n_iters=2, pose_damping=1e-5, pose_ep=1e-2, motion_only=False, limited_disp=False,
              optimize_intrinsics=intr)
    p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole", bk)
    E = len(g.ii)
    op, od, ok_, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                           ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                           g.eta[:, None], g.ii, g.jj, **bk)
    assert info[0] == 69 and info[3] == 414 + int(intr) and info[2] == 0 and info[5] == 0
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max()), np.abs(p - op).max()
    assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()
    assert np.abs(k - ok_).max() <= 1e-4 * np.abs(ok_).max()
    assert np.abs(p - g.poses).max() > 1e-4


def test_dense_ba_non_banded_graph_takes_global_memory_solver():
    """Long-range (loop-closure-like) edges make the reduced system dense: it no longer fits the LDS band solver and
    the blocked global-memory Cholesky (fp64 MFMA trailing update) must give the same answer as the fp64 oracle."""
    g = make_graph(n=30, height=96, width=128, radius=2, extra_edges=60, seed=77)
    bk = dict(t0=1, t1=30, n_iters=2, pose_damping=1e-5, pose_ep=1e-2, motion_only=False, limited_disp=False,
              optimize_intrinsics=True)
    p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole", bk)
    E = len(g.ii)
    op, od, ok_, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                           ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                           g.eta[:, None], g.ii, g.jj, **bk)
    assert info[0] == 29 and info[3] == 175 and info[2] == 0
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()
    assert np.abs(k - ok_).max() <= 1e-4 * np.abs(ok_).max()


@pytest.mark.parametrize("shape", [(3, 48, 64), (1, 16, 128)])
def test_corr_pyramid_build_fused_kernel(shape):
    """Row a1 (droid_net.py:56-69,94-102): volume + pooled pyramid in one HIP kernel.  Level 0 against the fp32
    contraction of the same fp16 inputs (one rounding to half; the summation order of an fp32-accumulating GEMM is
    not pinned by the reference either), levels 1..3 BIT-EXACT against avg_pool2d's arithmetic (fp32 sum of the
    window in row-major order, /4, round to half) applied to the kernel's own previous level, and all levels against
    the library path (hipBLASLt matmul + at::avg_pool2d) the reference would run."""
    from vipe_amd.ext import droid_net_ext

    E, h, w = shape
    g = torch.Generator().manual_seed(5)
    f1 = torch.randn(E, 128, h, w, generator=g).half().to(dev())
    f2 = torch.randn(E, 128, h, w, generator=g).half().to(dev())
    lv = droid_net_ext.corr_pyramid_build(f1, f2, 4)
    assert [tuple(x.shape) for x in lv] == [(E, h, w, h >> i, w >> i) for i in range(4)]
    ref0 = torch.matmul((f1.float() / 4).reshape(E, 128, h * w).transpose(1, 2), (f2.float() / 4).reshape(E, 128, h * w))
    d0 = (lv[0].float().reshape(E, h * w, h * w) - ref0).abs()
    assert float((d0 - ref0.abs() * 2.0 ** -10).max()) <= 2.0 ** -14, "level 0 is not the rounded fp32 contraction"
    for i in range(3):
        x = lv[i].float().reshape(-1, h >> i, w >> i)
        pooled = (((x[:, 0::2, 0::2] + x[:, 0::2, 1::2]) + x[:, 1::2, 0::2]) + x[:, 1::2, 1::2]) / 4.0
        assert torch.equal(pooled.half().reshape(lv[i + 1].shape), lv[i + 1]), f"level {i + 1} pooling not bit-exact"
    vol = torch.matmul((f1 / 4.0).reshape(E, 128, h * w).transpose(1, 2), (f2 / 4.0).reshape(E, 128, h * w))  # hipBLASLt
    vol = vol.reshape(E * h * w, 1, h, w)
    for i in range(4):
        assert float((vol.view(lv[i].shape).float() - lv[i].float()).abs().max()) <= 2e-2
        if i < 3:
            vol = torch.nn.functional.avg_pool2d(vol, 2, stride=2)


def test_add_proximity_factors_on_device_buffer():
    """Edge proposal end to end on the device buffer (frame_distance kernel -> host NMS, factor_graph.py:411-488):
    the edge list equals the one the same selection logic produces from the ORACLE's frame distances (the selection
    logic itself is pinned against the reference's own code in test_oracle_golden)."""
    import types

    from oracle import frame_ops as ofo
    from vipe_amd.slam.factor_graph import FactorGraph

    g, buf, graph, um = _tiny_graph(n=7, seed=44)
    n = 7
    graph.rm_factors(torch.ones(len(graph.ii), dtype=torch.bool))  # start from an empty graph
    graph.max_factors = 64
    graph.add_proximity_factors(t0=0, t1=0, rad=1, nms=1, beta=0.25, thresh=1e3, remove=False)
    got = np.stack([graph.ii.cpu().numpy(), graph.jj.cpu().numpy()], 1)
    # oracle distances into the same host logic
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    ii, jj = ii.reshape(-1), jj.reshape(-1)
    z = np.zeros_like(ii)
    intr8 = (g.intrinsics / 8.0).astype(np.float32)
    dij = ofo.frame_distance(g.poses, g.disps, intr8, ii, jj, z, z, ii, 0.25)
    dji = ofo.frame_distance(g.poses, g.disps, intr8, jj, ii, z, z, jj, 0.25)
    D = torch.from_numpy((0.5 * (dij + dji)).reshape(n, n))  # buffer.py:585-590 bidirectional mean
    ref = object.__new__(FactorGraph)
    ref.buffer = types.SimpleNamespace(n_frames=n, n_views=1, frame_distance_dense_disp=lambda a, b, beta: D[a, b][:, None])
    ref.device, ref.cross_view, ref.max_factors = torch.device("cpu"), False, 64
    ref.ii = ref.jj = ref.ii_inac = ref.jj_inac = torch.zeros(0, dtype=torch.long)
    out = {}
    ref.add_factors = lambda a, b, remove=False: out.update(e=np.stack([np.asarray(a), np.asarray(b)], 1))
    ref.add_proximity_factors(t0=0, t1=0, rad=1, nms=1, beta=0.25, thresh=1e3, remove=False)
    assert len(got) > 2 * (n - 1) and np.array_equal(got, out["e"])


def test_config5_grid_dense_ba_and_update_operator():
    """BASELINE config 5 shape (1024x512 image -> 64x128 grid, P = 8192, pinhole model - the reference has no panorama
    projection): a small graph through the dense BA against the fp64 oracle (1e-4), and the flow-update operator on
    the halo-tile convolution path for width 128 against the torch-fp32 restatement of the reference operator."""
    from oracle import update_module as oum
    from vipe_amd.slam.networks import UpdateModule

    g = make_graph(n=3, height=512, width=1024, radius=2, seed=55)
    assert (g.ht, g.wd) == (64, 128)
    bk = dict(t0=1, t1=3, n_iters=2, pose_damping=1e-3, pose_ep=0.1, motion_only=False, limited_disp=False,
              optimize_intrinsics=False)
    p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole", bk)
    E = len(g.ii)
    op, od, _, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                         ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                         g.eta[:, None], g.ii, g.jj, **bk)
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()
    # flow-update operator at 64 x 128
    torch.manual_seed(0)
    um = UpdateModule().eval()
    gen = torch.Generator().manual_seed(8)
    n = 2
    net = torch.randn(1, n, 128, 64, 128, generator=gen).tanh().half()
    inp = torch.randn(1, n, 128, 64, 128, generator=gen).relu().half()
    corr = (torch.randn(1, n, 196, 64, 128, generator=gen) * 0.5).half()
    flow = (torch.randn(1, n, 4, 64, 128, generator=gen) * 2).half()
    ix = torch.tensor([0, 1])
    eng = um.engine(dev())
    net_d, delta_d, weight_d, eta_d, _ = eng.forward(net.to(dev()), inp.to(dev()), corr.to(dev()), flow.to(dev()), ix.to(dev()),
                                                     skip_upmask=True)
    sd = {kk: w.float() for kk, w in um.state_dict().items()}
    with torch.no_grad():
        net_r, delta_r, weight_r, eta_r, _ = oum.update_forward(sd, net.float(), inp.float(), corr.float(), flow.float(), ix)
    assert (net_d.float().cpu() - net_r).abs().max().item() < 0.03
    assert (delta_d.float().cpu() - delta_r).abs().max().item() < 0.05
    assert (weight_d.float().cpu() - weight_r).abs().max().item() < 0.02
    assert (eta_d.float().cpu() - eta_r).abs().max().item() < 2e-3


def test_fused_lookup_conv1x1_matches_lookup_then_conv():
    """`vipe_corr_lookup_conv1x1` = lookup_nhwc (bit-exact, tested above) followed by the 1x1 convolution: the fused
    kernel must agree with the two-kernel composition to fp16 rounding of the output (same fp16 inputs, fp32
    accumulation in a different order), incl. a ragged last pixel group and out-of-range windows."""
    from vipe_amd._lib import check, lib, ptr, stream_ptr
    from vipe_amd.ext import droid_net_ext
    from vipe_amd.slam.update_engine import _Packed

    E, h, w = 2, 24, 64  # level 3 is 3 x 8
    gen = torch.Generator().manual_seed(12)
    f1 = torch.randn(E, 128, h, w, generator=gen).half().to(dev())
    f2 = torch.randn(E, 128, h, w, generator=gen).half().to(dev())
    lv = droid_net_ext.corr_pyramid_build(f1, f2, 4)
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    base = torch.from_numpy(np.stack([u, v], -1))[None].repeat(E, 1, 1, 1)
    coords = (base + 6.0 * torch.randn(E, h, w, 2, generator=gen)).to(dev()).contiguous()
    coords[0, 0, :8] = -50.0  # windows completely outside
    wt = torch.zeros(128, 200, 1, 1)
    wt[:, :196] = torch.randn(128, 196, 1, 1, generator=gen) / 14.0
    bias = torch.randn(128, generator=gen) * 0.1
    pk = _Packed(wt.half(), bias, dev())
    corr = droid_net_ext.corr_pyramid_lookup_nhwc(lv, coords, 3, 200)
    ref = torch.empty(E, h, w, 128, dtype=torch.float16, device=dev())
    check(lib().vipe_conv2d_nhwc_f16(ptr(corr), ptr(pk.packed), ptr(pk.bias), None, ptr(ref), E, h, w, 200, 200, 0, 128,
                                     128, 0, 1, 1, 1, stream_ptr(corr)), "conv")
    out = torch.full((E, h, w, 160), 3.0, dtype=torch.float16, device=dev())
    droid_net_ext.corr_lookup_conv1x1(lv, coords, pk.packed, pk.bias, out, out_coff=16, act="relu")
    assert float((out[..., 16:144].float() - ref.float()).abs().max()) <= 2e-2 * max(1.0, float(ref.float().abs().max()))
    assert torch.all(out[..., :16] == 3.0) and torch.all(out[..., 144:] == 3.0)
    # against the fp32 restatement: relu(W corr + b)
    r32 = torch.relu(torch.einsum("ehwc,oc->ehwo", corr[..., :196].float().cpu(), wt[:, :196, 0, 0].half().float()) + bias)
    assert float((out[..., 16:144].float().cpu() - r32).abs().max()) < 3e-2


def test_natively_sequenced_operator_equals_python_sequenced():
    """`vipe_update_operator` (one library call for the 13 convolutions, the lookup, the segmented mean and the pooled
    context product) issues exactly the kernels `forward_nhwc(native=False)` issues one by one: bit-identical state,
    heads and eta, with and without the hoisted gate context, from a deferred lookup and from materialised features."""
    from vipe_amd.slam.networks import CorrBlock, UpdateModule
    from vipe_amd.slam.update_engine import segment_csr
    torch.manual_seed(0)
    eng = UpdateModule().eval().engine(dev())
    E, h, w = 5, 8, 64
    g = torch.Generator().manual_seed(3)
    net = torch.randn(E, h, w, 128, generator=g).tanh().half().to(dev())
    xbuf = torch.zeros(E, h, w, 320, dtype=torch.float16, device=dev())
    xbuf[..., :128] = torch.randn(E, h, w, 128, generator=g).relu().half().to(dev())
    motn = (torch.randn(E, h, w, 4, generator=g) * 3).half().to(dev())
    fm = torch.randn(4, 128, h, w, generator=g).half().to(dev())
    i1, i2 = torch.tensor([0, 1, 2, 3, 0], device=dev()), torch.tensor([1, 2, 3, 0, 2], device=dev())
    cb = CorrBlock.from_buffer(fm, i1, i2)
    coords = (torch.rand(E, h, w, 2, generator=g) * torch.tensor([w - 1.0, h - 1.0])).to(dev())
    ix = torch.tensor([0, 0, 1, 2, 2], device=dev())
    csr = segment_csr(ix, 3)
    pg = eng.gate_context(xbuf)
    for corr in (cb.lookup_deferred(coords), cb.lookup_nhwc(coords)):
        for pgate in (pg, None):
            outs = []
            for native in (True, False):
                xb = xbuf.clone()
                n2, dw, eta, _ = eng.forward_nhwc(net, xb, corr, motn, ix=ix, n_src=3, csr=csr, pgate=pgate, native=native)
                outs.append((n2.clone(), dw.clone(), eta.clone(), xb))
            for a, b_ in zip(*outs):
                assert torch.equal(a, b_)
    # without GraphAgg (ix None), and the upmask branch on top of the native call
    n2a, dwa, eta_a, up_a = eng.forward_nhwc(net, xbuf.clone(), cb.lookup_deferred(coords), motn, native=True)
    n2b, dwb, _, _ = eng.forward_nhwc(net, xbuf.clone(), cb.lookup_deferred(coords), motn, native=False)
    assert eta_a is None and up_a is None and torch.equal(n2a, n2b) and torch.equal(dwa.clone(), dwb)
    _, _, _, up1 = eng.forward_nhwc(net, xbuf.clone(), cb.lookup_deferred(coords), motn, ix=ix, n_src=3, csr=csr, want_upmask=True)
    _, _, _, up2 = eng.forward_nhwc(net, xbuf.clone(), cb.lookup_deferred(coords), motn, ix=ix, n_src=3, csr=csr, want_upmask=True,
                                    native=False)
    assert up1.shape == (3, h, w, 576) and torch.equal(up1, up2)


def test_gate_context_hoisting_equals_full_gate_convolutions():
    """`UpdateEngine.gate_context` + accumulator initialisation (linearity of the gate convolutions in their input
    channels) against the full 448-channel gate convolutions on the same inputs: identical up to the fp16 rounding of
    the hoisted partial sum (the reference rounds once, after the full fp32 accumulation)."""
    from vipe_amd.slam.networks import UpdateModule

    torch.manual_seed(0)
    um = UpdateModule().eval()
    eng = um.engine(dev())
    gen = torch.Generator().manual_seed(21)
    E, H, W = 3, 8, 64
    net = torch.randn(E, H, W, 128, generator=gen).tanh().half().to(dev())
    xbuf = torch.zeros(E, H, W, 320, dtype=torch.float16, device=dev())
    xbuf[..., :128] = torch.randn(E, H, W, 128, generator=gen).relu().half().to(dev())
    corr = torch.zeros(E, H, W, 200, dtype=torch.float16, device=dev())
    corr[..., :196] = (torch.randn(E, H, W, 196, generator=gen) * 0.5).half().to(dev())
    motn = (torch.randn(E, H, W, 4, generator=gen) * 2).half().to(dev())
    ix = torch.tensor([0, 0, 1], device=dev())
    outs = []
    for hoist in (False, True):
        xb = xbuf.clone()
        pg = eng.gate_context(xb) if hoist else None
        n2, dw, eta, _ = eng.forward_nhwc(net.clone(), xb, corr, motn, ix=ix, n_src=2, pgate=pg)
        outs.append((n2.float().cpu(), dw.clone().cpu(), eta.clone().cpu()))
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 4e-3
    assert (outs[0][1] - outs[1][1]).abs().max().item() < 2e-2
    assert (outs[0][2] - outs[1][2]).abs().max().item() < 1e-3


def test_two_stream_operator_equals_single_stream():
    """`vipe_update_buffers.side_stream`: the flow encoder next to the lookup + correlation encoder and the heads next to
    the GraphAgg chain on a second stream, forked / joined with events inside the call - the same kernels on the same
    data, so everything but the atomically pooled global-context sum is bit-identical (the hidden state inherits its
    last-bit noise through the gates)."""
    from vipe_amd.slam.networks import CorrBlock, UpdateModule
    from vipe_amd.slam.update_engine import segment_csr
    torch.manual_seed(0)
    eng = UpdateModule().eval().engine(dev())
    E, h, w = 6, 8, 64
    g = torch.Generator().manual_seed(5)
    net = torch.randn(E, h, w, 128, generator=g).tanh().half().to(dev())
    xbuf = torch.zeros(E, h, w, 320, dtype=torch.float16, device=dev())
    xbuf[..., :128] = torch.randn(E, h, w, 128, generator=g).relu().half().to(dev())
    motn = (torch.randn(E, h, w, 4, generator=g) * 3).half().to(dev())
    fm = torch.randn(4, 128, h, w, generator=g).half().to(dev())
    i1, i2 = torch.tensor([0, 1, 2, 3, 0, 1], device=dev()), torch.tensor([1, 2, 3, 0, 2, 3], device=dev())
    cb = CorrBlock.from_buffer(fm, i1, i2)
    coords = (torch.rand(E, h, w, 2, generator=g) * torch.tensor([w - 1.0, h - 1.0])).to(dev())
    ix = torch.tensor([0, 0, 1, 2, 2, 1], device=dev())
    csr = segment_csr(ix, 3)
    pg = eng.gate_context(xbuf)
    outs = []
    for min_edges in (1, 10 ** 9):
        eng.op_side_min_edges = min_edges
        eng._bdesc.clear()
        xb = xbuf.clone()
        n2, dw, eta, _ = eng.forward_nhwc(net, xb, cb.lookup_deferred(coords), motn, ix=ix, n_src=3, csr=csr, pgate=pg)
        torch.cuda.synchronize()
        outs.append((n2.clone(), dw.clone(), eta.clone(), xb))
    assert eng._op_side is not None
    assert torch.equal(outs[0][3], outs[1][3])  # encoders' outputs: bit-identical
    assert (outs[0][0].float() - outs[1][0].float()).abs().max().item() < 2e-3
    assert (outs[0][1] - outs[1][1]).abs().max().item() < 1e-2
    assert (outs[0][2] - outs[1][2]).abs().max().item() < 1e-4


def test_staged_hidden_gate_state_equals_unsplit_gates():
    """`UpdateEngine.hidden_gate_state` (global-context terms + the hidden-state third of the z|r convolution as raw
    fp32 accumulators, VIPE_CONV_PARTIAL) followed by the operator with `gate_state=` (z|r over the corr | flow
    channels, accumulators starting from the fp32 partial sums) against the unsplit operator: the same sums in another
    fp32 order, so new state, heads and eta agree to fp16 rounding of a last-bit difference; natively sequenced and
    Python sequenced forms are bit-identical; a stale gate state (other hidden state) is ignored."""
    from vipe_amd.slam.networks import UpdateModule
    from vipe_amd.slam.update_engine import segment_csr

    torch.manual_seed(0)
    eng = UpdateModule().eval().engine(dev())
    gen = torch.Generator().manual_seed(33)
    E, H, W = 5, 8, 64
    net = torch.randn(E, H, W, 128, generator=gen).tanh().half().to(dev())
    xbuf = torch.zeros(E, H, W, 320, dtype=torch.float16, device=dev())
    xbuf[..., :128] = torch.randn(E, H, W, 128, generator=gen).relu().half().to(dev())
    corr = torch.zeros(E, H, W, 200, dtype=torch.float16, device=dev())
    corr[..., :196] = (torch.randn(E, H, W, 196, generator=gen) * 0.5).half().to(dev())
    motn = (torch.randn(E, H, W, 4, generator=gen) * 2).half().to(dev())
    ix = torch.tensor([0, 0, 1, 2, 2], device=dev())
    csr = segment_csr(ix, 3)
    pg = eng.gate_context(xbuf)
    ref = eng.forward_nhwc(net, xbuf.clone(), corr, motn, ix=ix, n_src=3, csr=csr, pgate=pg)
    ref = [t.clone() for t in ref[:3]]
    outs = {}
    for native in (True, False):
        gs = eng.hidden_gate_state(net, pg, native=native)
        assert eng.gate_state_matches(gs, net, pg)
        pzr = gs["pzr"].clone()
        o = eng.forward_nhwc(net, xbuf.clone(), corr, motn, ix=ix, n_src=3, csr=csr, pgate=pg, gate_state=gs, native=native)
        outs[native] = [t.clone() for t in o[:3]] + [pzr]
    for a, b_ in zip(outs[True], outs[False]):
        assert torch.equal(a, b_)
    # staged for the first 2 of the 5 edges only: those edges as above, the others bit-identical to the unsplit operator
    part = {}
    for native in (True, False):
        gs = eng.hidden_gate_state(net, pg, native=native, n_staged=2)
        assert gs["n_staged"] == 2 and gs["pzr"].shape[0] == 2
        o = eng.forward_nhwc(net, xbuf.clone(), corr, motn, ix=ix, n_src=3, csr=csr, pgate=pg, gate_state=gs, native=native)
        part[native] = [t.clone() for t in o[:3]]
    for a, b_ in zip(part[True], part[False]):
        assert torch.equal(a, b_)
    assert torch.equal(part[True][0][:2], outs[True][0][:2]) and torch.equal(part[True][0][2:], ref[0][2:])
    assert torch.equal(part[True][1][:2], outs[True][1][:2]) and torch.equal(part[True][1][2:], ref[1][2:])
    # the partial sums themselves: fp32 conv of the hidden state + the hoisted context part
    w = torch.cat([eng.m.gru.convz.weight, eng.m.gru.convr.weight], 0)[:, 0:128].detach().half().float().to(dev())
    want = torch.nn.functional.conv2d(net.float().permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1) + pg[..., :256].float()
    assert (outs[True][3] - want).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item())
    assert (outs[True][0].float() - ref[0].float()).abs().max().item() < 2e-3   # new hidden state (fp16 ulp at |x| <= 1)
    assert (outs[True][1] - ref[1]).abs().max().item() < 1e-2                   # heads (delta in pixels, fp16)
    assert (outs[True][2] - ref[2]).abs().max().item() < 1e-4                   # eta
    # a gate state of another hidden state must not be used
    gs = eng.hidden_gate_state(net, pg)
    other = net.flip(0).contiguous()
    a = eng.forward_nhwc(other, xbuf.clone(), corr, motn, ix=ix, n_src=3, csr=csr, pgate=pg, gate_state=gs)
    a = [t.clone() for t in a[:3]]
    b_ = eng.forward_nhwc(other, xbuf.clone(), corr, motn, ix=ix, n_src=3, csr=csr, pgate=pg)
    for x, y in zip(a, b_[:3]):
        assert torch.equal(x, y)


def test_update_with_gate_state_on_side_stream_matches_serial_update():
    """`FactorGraph.update` with the next iteration's hidden-state gate part issued on a second stream under the BA
    (the default for >= 64 active edges) against the same three iterations without it: same poses / disparities /
    targets up to the fp32 summation order of the split z|r convolution; edge-set changes drop the staged state."""
    import bench
    res = {}
    for overlap in ("gated", "free", False):
        g, buf, graph = bench.build_problem(dev(), 10, 128, 512, 3, 0, seed=77)
        graph.gate_overlap_min_edges = 1 if overlap else 10 ** 9
        graph.gate_overlap_mode = overlap or "gated"
        for _ in range(3):
            graph.update(t0=1, t1=10, itrs=2)
        assert (graph._gate_state is not None) == bool(overlap)
        torch.cuda.synchronize()
        res[overlap] = (buf.poses[:10].clone(), buf.disps[:10].clone(), graph.target.clone(), graph.net_n.clone())
        if overlap:  # removing edges invalidates the staged state; the next update recomputes in line and stages anew
            graph.rm_factors(graph.ii == 0, store=False)
            assert graph._gate_state is None
            graph.update(t0=1, t1=10, itrs=2)
            assert graph._gate_state is not None and torch.isfinite(buf.poses[:10]).all()
    # the same kernels on the same data, released differently (not bit-identical: the pooled global-context sum is
    # accumulated with float atomics in workgroup order)
    assert (res["gated"][0] - res["free"][0]).abs().max().item() < 1e-5
    assert (res["gated"][3].float() - res["free"][3].float()).abs().max().item() < 2e-3
    assert (res["gated"][3].float() - res[False][3].float()).abs().max().item() < 1e-2
    assert (res["gated"][2] - res[False][2]).abs().max().item() < 5e-2
    assert (res["gated"][0] - res[False][0]).abs().max().item() < 1e-4
    assert (res["gated"][1] - res[False][1]).abs().max().item() < 1e-4


@pytest.mark.parametrize("case", ["full", "motion_only", "prior_stereo_t0"])
def test_slam_ext_ba_droid_signature_against_restatement(case):
    """Row a13: `slam_ext.ba` (DROID signature, geom_kernels.cu:1273-1404; dormant in the reference).  PARITY UNPINNED:
    the reference's CUDA/Eigen implementation cannot run here and has no fixture; the check is the HIP path against
    the CPU restatement of the source text (oracle/droid_ba.py, incl. its quirks), 1e-4 relative."""
    from oracle import droid_ba as odb
    from vipe_amd.ext import slam_ext

    g = make_graph(n=6, height=96, width=128, radius=2, seed=61, depth_prior=(case == "prior_stereo_t0"))
    n, ht, wd = 6, g.ht, g.wd
    ii, jj = g.ii.copy(), g.jj.copy()
    tgt = g.target.reshape(len(ii), ht, wd, 2)
    wgt = g.weight.reshape(len(ii), ht, wd, 2)
    t0, t1 = (2, 6) if case == "prior_stereo_t0" else (1, 6)
    if case == "prior_stereo_t0":  # a stereo term (ii == jj): only its disparity part may act
        ii, jj = np.append(ii, 3), np.append(jj, 3)
        u, v = ogeom.pixel_grid(ht, wd, np.float32)
        st = np.stack([u - 0.1 * g.intrinsics[0, 0] / 8 * g.disps[3], v], -1)[None]
        tgt = np.concatenate([tgt, st + 0.3], 0)
        wgt = np.concatenate([wgt, np.full_like(st, 0.5)], 0)
    targets = np.ascontiguousarray(tgt.transpose(0, 3, 1, 2)).astype(np.float32)
    weights = np.ascontiguousarray(wgt.transpose(0, 3, 1, 2)).astype(np.float32)
    kx = np.unique(np.concatenate([np.arange(t0, t1), ii]))
    rng = np.random.default_rng(4)
    eta = (1e-3 + 0.05 * rng.random((len(kx), ht, wd))).astype(np.float32)
    intr8 = (g.intrinsics[0, :4] / 8.0).astype(np.float32)
    sens = g.disps_sens.copy()
    if case == "prior_stereo_t0":
        sens[:, ::2] = 0.0  # per-pixel mask: half of the rows have no sensor depth
    mo = case == "motion_only"
    poses, disps = T(g.poses).clone(), T(g.disps).clone()
    dx, dz = slam_ext.ba(poses, disps, T(intr8), T(sens), T(targets), T(weights), T(eta), T(ii), T(jj), t0, t1, 2,
                         1e-4, 0.1, mo)
    op, od, odx, odz = odb.droid_ba(g.poses, g.disps, intr8, sens, targets, weights, eta, ii, jj, t0, t1, 2, 1e-4, 0.1, mo)
    assert np.abs(poses.cpu().numpy() - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(disps.cpu().numpy() - od).max() <= 1e-4 * np.abs(od).max()
    assert np.abs(dx.cpu().numpy() - odx).max() <= 1e-4 * max(np.abs(odx).max(), 1e-2)
    if not mo:
        assert np.abs(dz.cpu().numpy() - odz).max() <= 1e-4 * max(np.abs(odz).max(), 1e-2)
        assert np.abs(odz).max() > 1e-4
    else:
        assert np.all(disps.cpu().numpy() == g.disps)


def test_altcorr_backward_is_adjoint_of_forward():
    """`altcorr_backward` (altcorr_kernel.cu:140-264, 292-320; float32): the forward is bilinear in (fmap1, fmap2), so
    <corr_grad, altcorr(f1 + e d1, f2 + e d2)>' at e = 0 must equal <g1, d1> + <g2, d2> (exactly, up to fp32 rounding)."""
    from vipe_amd.ext import droid_net_ext

    gen = torch.Generator().manual_seed(17)
    B, H, W, C, N = 2, 8, 10, 64, 3
    f1 = torch.randn(B, H, W, C, generator=gen).to(dev())
    f2 = torch.randn(B, H, W, C, generator=gen).to(dev())
    u, v = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    coords = (torch.from_numpy(np.stack([u, v], -1))[None, None] + 3 * torch.randn(B, N, H, W, 2, generator=gen)).to(dev()).contiguous()
    cg = torch.randn(B, N, 49, H, W, generator=gen).to(dev())
    g1, g2, gc = droid_net_ext.altcorr_backward(f1, f2, coords, cg, 3)
    d1 = torch.randn(B, H, W, C, generator=gen).to(dev())
    d2 = torch.randn(B, H, W, C, generator=gen).to(dev())
    fwd = lambda a, b: droid_net_ext.altcorr_forward(a, b, coords, 3)[0]
    lhs = (cg * (fwd(d1, f2) + fwd(f1, d2))).double().sum()  # directional derivative of a bilinear map
    rhs = (g1 * d1).double().sum() + (g2 * d2).double().sum()
    assert abs(float(lhs - rhs)) <= 1e-4 * max(1.0, abs(float(lhs)))
    assert float(gc.abs().max()) == 0.0


def test_dense_ba_medium_degree_uses_large_mfma_accumulate():
    """Source frames with 7..14 terms (radius-5 graph: degree 10) take the large-LDS instantiation of the matrix-core
    accumulate kernel; the reduced system is wider than the LDS band solver's limit, so the global-memory Cholesky
    solves it.  Against the fp64 oracle, 1e-4 (incl. focal optimisation)."""
    g = make_graph(n=14, height=96, width=128, radius=5, seed=88)
    deg = np.bincount(g.ii).max()
    assert 7 <= deg <= 14
    bk = dict(t0=1, t1=14, n_iters=2, pose_damping=1e-3, pose_ep=0.1, motion_only=False, limited_disp=False,
              optimize_intrinsics=True)
    p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole", bk)
    E = len(g.ii)
    op, od, ok_, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                           ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                           g.eta[:, None], g.ii, g.jj, **bk)
    assert info[2] == 0
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()
    assert np.abs(k - ok_).max() <= 1e-4 * np.abs(ok_).max()


def test_frontend_mirror_runs_and_keeps_graph_state_consistent():
    """`SLAMFrontend` (frontend.py:78-167 mirror) on 14 synthetic keyframes of a 128x128 clip: initialisation after the
    warm-up, proximity edges, update iterations with inactive edges, pose extrapolation.  Checks the bookkeeping
    invariants the reference's tensors obey (one volume / hidden state / target per edge and view, ages advance, the
    window holds at most max_factors edges after a removal round) and that the state stays finite."""
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.frontend import FrontendArgs, SLAMFrontend
    from vipe_amd.slam.networks import UpdateModule

    torch.manual_seed(0)
    um = UpdateModule().eval()
    buf = GraphBuffer(128, 128, buffer_size=24, device=dev())
    buf.intrinsics[:] = torch.tensor([115.0, 115.0, 64.0, 64.0], device=dev())
    fe = SLAMFrontend(um, buf, FrontendArgs(keyframe_thresh=0.0), dev())
    gen = torch.Generator().manual_seed(7)
    for t in range(14):
        buf.fmaps[t, 0] = torch.randn(128, 16, 16, generator=gen).half().to(dev())
        buf.nets[t, 0] = torch.randn(128, 16, 16, generator=gen).tanh().half().to(dev())
        buf.inps[t, 0] = torch.randn(128, 16, 16, generator=gen).relu().half().to(dev())
        if t < 8:
            buf.poses[t, 0] = 0.05 * t
            buf.disps[t, 0] = (1.0 / (1.0 + 4.0 * torch.rand(16, 16, generator=gen))).to(dev())
        buf.n_frames += 1
        fe.run()
    g = fe.graph
    E = int(g.ii.numel())
    assert fe.is_initialized and fe.t1 == 14 and fe.n_updates == 8 + 6 * 6
    assert 0 < E <= 48 + 2 * 3 and g.jj.numel() == E and g.age.numel() == E
    assert g.corr.corr_pyramid[0].shape[0] == E and g.net_n.shape[0] == E and g.xbuf.shape[0] == E
    assert g.target.shape[1] == E and g.weight.shape[1] == E
    assert g.ii_inac.numel() == g.target_inac.shape[1] and g.ii_inac.numel() > 0
    assert int(g.age.max()) >= 6 and int(g.ii.max()) == 13
    assert bool(torch.isfinite(buf.poses[:15]).all()) and bool(torch.isfinite(buf.disps[:15]).all())
    assert bool((buf.disps[:14] >= 1e-3).all())


def test_clip_pipeline_at_bench_resolution_from_rgb():
    """BASELINE configs[1]-shaped run at the bench resolution (512 x 384, 48 x 64 grid - the shapes every fast kernel
    is specialised for) through the product's entry point, exactly as `bench.py --mode video` runs it: ONE
    `SLAMSystem.run(frames)` - RGB + sensor-depth frames -> motion filter (feature / context encoders, one operator
    application; `SLAMConfig.pipeline_filter`: prefetched on the side stream under the previous keyframe's frontend
    step) -> keyframe frontend (proximity edges, pyramids into the pooled store, 4 + 2 update iterations with inactive
    edges in the BA) -> the two global-BA passes -> pass 2.  The pipelined and the serial schedule must agree on the
    bookkeeping and on the filter's scores; the pipelined run must really have collected prefetched filter work."""
    import bench
    from vipe_amd.slam.motion_filter import MotionFilter
    res, collected = {}, {True: 0, False: 0}
    real_finish = MotionFilter.finish
    for pipelined in (True, False):
        def finish(self, h, _p=pipelined):
            collected[_p] += int(h["stream"] is not None)
            return real_finish(self, h)
        MotionFilter.finish = finish
        try:
            run_clip = bench.make_clip_runner(dev(), pipelined=pipelined)
            res[pipelined] = run_clip(seed=3, n_frames=14)
        finally:
            MotionFilter.finish = real_finish
    assert collected[True] == 13 and collected[False] == 0  # every frame but the first came through the side stream
    for r in res.values():
        assert r["finite"] and r["keyframes"] == 14 and r["update_iterations"] == 8 + 6 * 6
        assert tuple(r["poses"].shape) == (14, 7)  # pass 2: every frame got a pose (here all of them are keyframes)
        assert 0 < r["edges_final"] <= 48 + 2 * 3 and r["backend_edges"] > r["edges_final"]
        q = r["poses"][:, 3:]
        assert (q.norm(dim=-1) - 1.0).abs().max().item() < 1e-4  # unit quaternions after every retraction
        assert r["pass1_seconds"] < r["seconds_to_global_ba_done"] < r["seconds_to_pass2_done"] <= r["seconds"]
    assert res[True]["edges_final"] == res[False]["edges_final"] and res[True]["backend_edges"] == res[False]["backend_edges"]
    # The filter's score depends on the images alone (frame f against the last keyframe), not on the SLAM state: the side
    # stream must reproduce the serial schedule's scores up to the atomically accumulated instance-norm / pooling sums.
    # The trajectories themselves are not compared closely: with random weights the recurrent iterations amplify those
    # last-bit differences from run to run (no checkpoint offline), whatever the schedule.
    sa, sb = np.array(res[True]["filter_scores"]), np.array(res[False]["filter_scores"])
    assert sa.shape == sb.shape == (13,) and np.abs(sa - sb).max() <= 5e-3 * np.abs(sb).max()
    # a scripted keep rate: every fourth frame becomes a keyframe, the others get their pose in pass 2
    r4 = bench.make_clip_runner(dev())(seed=4, n_frames=41, keep_every=4)
    assert r4["finite"] and tuple(r4["poses"].shape) == (41, 7) and r4["keyframes"] == 11  # 0, 4, ..., 40
    assert len(r4["filter_scores"]) == 40  # every frame after the first went through the whole filter


def test_backend_depth_prior_branch_with_a_pluggable_depth_model():
    """`SLAMBackend.run` with a depth model (backend.py:45-63): half of the passes with `optimize_intrinsics`, then
    `GraphBuffer.update_disps_sens` for the intrinsics found so far - the focal-ratio rescale for "metric_depth" models,
    per-frame re-estimation otherwise - then the rest with the intrinsics held.  The model is a stand-in (the reference's
    monocular depth networks are outside the path): what is checked is the control flow and the bookkeeping."""
    import bench
    from types import SimpleNamespace
    from vipe_amd.slam.backend import BackendArgs, SLAMBackend

    class FakeDepth:
        def __init__(self, kind):
            self.depth_type, self.calls, self.focals = kind, 0, []

        def estimate(self, inp):
            self.calls += 1
            self.focals.append(inp.focal_length)
            V, H, W, _ = inp.rgb.shape
            return SimpleNamespace(metric_depth=torch.full((V, H, W), 2.0, device=inp.rgb.device))

    for kind in ("metric_depth", "model_space"):
        g, buf, graph = bench.build_problem(dev(), 8, 128, 512, 3, 0, seed=31)
        buf.images[:8] = 0.5
        model = FakeDepth(kind)
        for f in range(8):  # what SLAMSystem does per keyframe (system.py:162)
            buf.update_disps_sens(model, frame_idx=f)
        assert model.calls == 8 and torch.allclose(buf.disps_sens[:8], torch.full_like(buf.disps_sens[:8], 0.5))
        f0 = buf.intrinsics[0, 0].item()
        be = SLAMBackend(graph.update_op, buf, BackendArgs(optimize_intrinsics=True), dev())
        be.depth_model = model
        gb = be.run(4)
        torch.cuda.synchronize()
        f1 = buf.intrinsics[0, 0].item()
        assert gb.ii.numel() > 0 and f1 != f0 and bool(torch.isfinite(buf.poses[:8]).all()) and bool(buf.dirty[:8].all())
        assert torch.allclose(buf.last_depth_intrinsics, buf.intrinsics) == (kind != "metric_depth")
        if kind == "metric_depth":   # no re-estimation: the prior is rescaled by the focal ratio at the half-way point
            assert model.calls == 8
            assert float((buf.disps_sens[:8] - 0.5).abs().max()) > 0 and float(buf.disps_sens[:8].std()) < 1e-6
        else:                        # re-estimated for all frames with the focal found by the first half
            assert model.calls == 16 and all(abs(x - f0) < 1e-6 for x in model.focals[:8]) and model.focals[8] != f0


def test_adaptive_cross_view_indices_on_a_two_view_rig():
    """`GraphBuffer.build_adaptive_cross_view_idx` (buffer.py:270-301): every (keyframe, view) gets the (keyframe, other
    view) of smallest one-directional reprojection distance when that is below the threshold, else keeps its partner -
    against the same selection written out with explicit loops over the distance tensor."""
    from vipe_amd.slam.buffer import GraphBuffer
    torch.manual_seed(5)
    n, V = 5, 2
    buf = GraphBuffer(96, 128, n_views=V, buffer_size=8, device=dev())
    buf.n_frames = n
    buf.intrinsics[:] = torch.tensor([100.0, 100.0, 64.0, 48.0], device=dev())
    buf.poses[:n, 0] = 0.1 * torch.arange(n, device=dev())
    buf.rig[1, 0] = 0.05
    buf.disps[:n] = 0.5 + 0.1 * torch.rand(n, V, 12, 16, device=dev())
    old = buf.cross_view_idx[:n].clone()
    ix = torch.arange(n, device=dev())
    ii, jj = [t.reshape(-1) for t in torch.meshgrid(ix, ix, indexing="ij")]
    d = buf.frame_distance_dense_disp(ii, jj, beta=1.0, view_offset=1, bidirectional=False).reshape(n, n, V).cpu()
    thresh = float(d.median())
    buf.build_adaptive_cross_view_idx(valid_thresh=thresh)
    got = buf.cross_view_idx[:n].cpu()
    for s in range(n):
        for v in range(V):
            t_best = int(torch.argmin(d[s, :, v]))
            if d[s, t_best, v] < thresh:
                assert got[s, v].tolist() == [t_best, (v + 1) % V]
            else:
                assert got[s, v].tolist() == old[s, v].cpu().tolist()


@pytest.mark.parametrize("dims,knn", [(2, 1), (3, 1), (3, 4), (2, 8)])
def test_nearest_neighbours_matches_brute_force(dims, knn):
    """`utils_ext.nearest_neighbours` (knn.cu:27-67: squared L2, [dist, idx] of shape [M, knn]) against torch.cdist +
    topk: identical distances; indices identical wherever the k-th and (k+1)-th distances differ."""
    from vipe_amd.ext import utils_ext
    g = torch.Generator().manual_seed(dims * 10 + knn)
    q = torch.rand(3001, dims, generator=g).to(dev())
    t = torch.rand(2500, dims, generator=g).to(dev())
    dist, idx = utils_ext.nearest_neighbours(q, t, knn)
    assert dist.shape == (3001, knn) and idx.shape == (3001, knn) and idx.dtype == torch.int32
    d2 = torch.cdist(q.double(), t.double()) ** 2
    want_d, want_i = torch.topk(d2, knn, dim=1, largest=False)
    assert (dist.double() - want_d).abs().max().item() < 1e-6
    got_d = torch.gather(d2, 1, idx.long())
    assert (got_d - want_d).abs().max().item() < 1e-12  # the returned indices realise the k smallest distances
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())
    with pytest.raises(RuntimeError):
        utils_ext.nearest_neighbours(q, t[:2], 4)  # knn > N (knn.cu:32)


def test_project_map_infill_takes_the_nearest_projected_point():
    """`SLAMMap.project_map(infill=True)` (interface.py:126-139): every pixel of the target view gets the depth of the
    projected map point nearest to its centre; pixels that contain a point agree with the non-infill rendering
    wherever that pixel holds exactly one point."""
    from vipe_amd.ext.lietorch import SE3
    from vipe_amd.slam.buffer import GraphBuffer
    torch.manual_seed(3)
    n = 4
    buf = GraphBuffer(96, 128, buffer_size=8, device=dev())
    buf.n_frames = n
    buf.intrinsics[:] = torch.tensor([100.0, 100.0, 64.0, 48.0], device=dev())
    buf.poses[:n, 0] = 0.02 * torch.arange(n, device=dev())
    buf.disps[:n] = 0.4 + 0.05 * torch.rand(n, 1, 12, 16, device=dev())
    buf.tstamp[:n] = torch.arange(n, device=dev(), dtype=torch.int)
    m = buf.extract_slam_map(filter_thresh=1e9)
    pose = SE3(buf.poses[1:2].clone()).inv()  # camera -> world of keyframe 1
    intr = buf.intrinsics[0] / 8.0
    sparse = m.project_map(1, 0, (12, 16), intr, pose[0])
    dense = m.project_map(1, 0, (12, 16), intr, pose[0], infill=True)
    assert dense.shape == (12, 16) and bool((dense > 0).all()) and bool(torch.isfinite(dense).all())
    hit = sparse > 0
    assert int(hit.sum()) > 20
    # a pixel's own point(s) are among the candidates: the nearest point's depth lies within the range of map depths
    assert float(dense.min()) >= float(sparse[hit].min()) - 1e-4 and float(dense.max()) <= float(sparse[hit].max()) + 0.5


def test_inner_filler_interpolates_and_refines_non_keyframe_poses():
    """`InnerFiller` (inner_filler.py:46-138, pass 2 of SLAMSystem.run): all 16 frames of a clip appended behind its 6
    keyframes.  The constant-velocity initial poses against the fp64 oracle group operations; then ten motion-only
    update iterations on the (keyframe -> frame) graph: keyframes untouched, frame count restored, finite refined poses
    that stay near their start (the synthetic targets are the graph's own reprojections)."""
    import bench
    from vipe_amd.slam.inner_filler import InfillArgs, InnerFiller

    n_kf, n_all = 6, 16
    g, buf, graph = bench.build_problem(dev(), n_kf, 128, 512, 3, 0, seed=41)
    kf_t = torch.tensor([0, 3, 6, 9, 12, 15], device=dev(), dtype=buf.tstamp.dtype)
    buf.tstamp[:n_kf] = kf_t
    kf_poses = buf.poses[:n_kf].clone()
    gen = torch.Generator().manual_seed(4)
    for f in range(n_all):  # pass 2 appends every frame of the video (system.py:286-292)
        t = buf.n_frames
        buf.tstamp[t] = f
        buf.fmaps[t, 0] = torch.randn(128, 16, 64, generator=gen).half().to(dev())
        buf.nets[t, 0] = torch.randn(128, 16, 64, generator=gen).tanh().half().to(dev())
        buf.inps[t, 0] = torch.randn(128, 16, 64, generator=gen).relu().half().to(dev())
        buf.n_frames += 1
    filler = InnerFiller(graph.update_op, buf, InfillArgs(infill_chunk_size=16), dev())
    filler.set_start_idx(n_kf)
    assert filler.check()
    t0, t1, m_pose = filler.interpolate()
    # oracle: Exp(Log(G_t1 G_t0^-1) * dt / DT) G_t0 in fp64
    P = kf_poses.double().cpu().numpy()
    t0n, t1n = t0.cpu().numpy(), t1.cpu().numpy()
    kt = kf_t.double().cpu().numpy()
    want = []
    for f in range(n_all):
        a, b = int(t0n[f]), int(t1n[f])
        assert a == min(f // 3, n_kf - 1) and b == min(a + 1, n_kf - 1)
        rel = ose3.se3_mul(P[b][None], ose3.se3_inv(P[a][None]))
        xi = ose3.se3_log(rel) * ((f - kt[a]) / (kt[b] - kt[a] + 1e-3))
        want.append(ose3.se3_mul(ose3.se3_exp(xi), P[a][None])[0])
    want = np.stack(want)
    got = m_pose.data.double().cpu().numpy()
    sign = np.sign((got[:, 3:] * want[:, 3:]).sum(-1, keepdims=True))  # q and -q are the same rotation
    assert np.abs(got[:, :3] - want[:, :3]).max() < 1e-5 and np.abs(got[:, 3:] * sign - want[:, 3:]).max() < 1e-5
    filler.compute()
    torch.cuda.synchronize()
    r = filler.get_result()
    assert buf.n_frames == n_kf and r.poses.data.shape == (n_all, 7) and r.dense_disps is None
    assert torch.equal(buf.poses[:n_kf], kf_poses) and bool(torch.isfinite(r.poses.data).all())
    assert (r.poses.data[:, 3:].norm(dim=-1) - 1.0).abs().max().item() < 1e-4
    assert int(filler.last_graph.ii.numel()) >= n_all  # every frame is tied to its neighbouring keyframe(s)


def test_frontend_prefetched_frame_distances_change_nothing():
    """The frontend launches the frame-distance kernel for the NEXT keyframe's edge proposal at the end of each step and
    reads the result from pinned memory one step later (no stream drain).  Same kernel, same inputs: the edge lists must
    be identical to a frontend that computes the distances on demand, keyframe after keyframe - and the prefetch must
    actually be hit, and be dropped when somebody touches the geometry in between."""
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.frontend import FrontendArgs, SLAMFrontend
    from vipe_amd.slam.networks import UpdateModule

    def make(prefetch):
        torch.manual_seed(0)
        buf = GraphBuffer(128, 128, buffer_size=24, device=dev())
        buf.intrinsics[:] = torch.tensor([115.0, 115.0, 64.0, 64.0], device=dev())
        fe = SLAMFrontend(UpdateModule().eval(), buf, FrontendArgs(keyframe_thresh=0.0), dev())
        if not prefetch:
            fe._prefetch_proximity = lambda: None
        hits = []
        orig = fe._prefetched_distances
        fe._prefetched_distances = lambda *a: (hits.append(1) or True) and (lambda r: (hits.pop() if r is None else None, r)[1])(orig(*a))
        return buf, fe, hits

    (b1, f1, h1), (b2, f2, h2) = make(True), make(False)
    gen = torch.Generator().manual_seed(7)
    for t in range(16):
        fm = torch.randn(128, 16, 16, generator=gen).half().to(dev())
        nt = torch.randn(128, 16, 16, generator=gen).tanh().half().to(dev())
        ip = torch.randn(128, 16, 16, generator=gen).relu().half().to(dev())
        dd = (1.0 / (1.0 + 4.0 * torch.rand(16, 16, generator=gen))).to(dev())
        for buf, fe in ((b1, f1), (b2, f2)):
            buf.fmaps[t, 0], buf.nets[t, 0], buf.inps[t, 0] = fm, nt, ip
            if t < 8:
                buf.poses[t, 0] = 0.05 * t
                buf.disps[t, 0] = dd
            if t == 12:
                buf.touch()  # an outside writer declares a change: this keyframe's prefetch must be discarded
            buf.n_frames += 1
            fe.run()
        assert torch.equal(f1.graph.ii, f2.graph.ii) and torch.equal(f1.graph.jj, f2.graph.jj), t
        assert torch.equal(f1.graph.ii_inac, f2.graph.ii_inac) and torch.equal(f1.graph.jj_inac, f2.graph.jj_inac), t
    assert len(h1) == 7 and len(h2) == 0  # keyframes 8..15 except the touched one
    assert float((b1.poses[:16] - b2.poses[:16]).abs().max()) < 1e-3  # same edges; float atomics differ run to run


def test_update_batch_volume_path_matches_altcorr_path(monkeypatch):
    """Hot loop B (factor_graph.py:316-394): on 288 GB parts the backend builds the per-chunk correlation volume and
    uses the fused lookup instead of the volume-free AltCorrBlock.  Same graph through both paths: the correlation
    features differ only by fp16 rounding of the volume (AltCorr accumulates the same dot products in fp32), so targets
    agree to a few hundredths of a pixel and the BA result to 1e-3."""
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.factor_graph import FactorGraph
    from vipe_amd.slam.networks import UpdateModule

    def run(altcorr):
        if altcorr:
            monkeypatch.setenv("VIPE_AMD_BACKEND_ALTCORR", "1")
        else:
            monkeypatch.delenv("VIPE_AMD_BACKEND_ALTCORR", raising=False)
        g = make_graph(n=4, height=64, width=512, radius=2, seed=52)  # 8 x 64 grid
        buf = GraphBuffer(64, 512, buffer_size=6, device=dev())
        buf.n_frames = 4
        buf.poses[:4], buf.disps[:4, 0], buf.intrinsics[:] = T(g.poses), T(g.disps), T(g.intrinsics)
        gen = torch.Generator().manual_seed(52)
        buf.fmaps[:4, 0] = torch.randn(4, 128, 8, 64, generator=gen).half().to(dev())
        buf.nets[:4, 0] = torch.randn(4, 128, 8, 64, generator=gen).tanh().half().to(dev())
        buf.inps[:4, 0] = torch.randn(4, 128, 8, 64, generator=gen).relu().half().to(dev())
        torch.manual_seed(0)
        graph = FactorGraph(UpdateModule().eval(), buf, dev(), max_factors=-1, incremental=False)
        graph.add_factors(torch.from_numpy(g.ii), torch.from_numpy(g.jj))
        graph.update_batch(itrs=2, steps=1, optimize_intrinsics=False, optimize_rig_rotation=False)
        torch.cuda.synchronize()
        return graph.target.cpu().numpy(), buf.poses[:4].cpu().numpy(), buf.disps[:4, 0].cpu().numpy()

    ta, pa, da = run(True)
    tv, pv, dv = run(False)
    assert np.abs(ta - tv).max() < 0.1
    assert np.abs(pa - pv).max() < 1e-3 and np.abs(da - dv).max() < 1e-2 * np.abs(da).max()


def test_extract_slam_map_and_project_map():
    """GraphBuffer.extract_slam_map (buffer.py:595-645) + SLAMMap (interface.py:25-141) against the oracle's
    depth_filter / iproj restatements; project_map keeps the nearest point per pixel and reproduces a keyframe's own
    depth when projected back into that keyframe."""
    from oracle import frame_ops
    from vipe_amd.ext.lietorch import SE3
    from vipe_amd.slam.buffer import GraphBuffer
    N, ht, wd = 6, 48, 64
    g = make_graph(N, ht * 8, wd * 8, seed=4)
    buf = GraphBuffer(ht * 8, wd * 8, buffer_size=8, device=dev())
    buf.poses[:N] = T(g.poses_gt, torch.float32)
    buf.disps[:N, 0] = T(g.disps_gt, torch.float32)
    buf.intrinsics[0] = T(g.intrinsics[0], torch.float32)
    buf.tstamp[:N] = torch.arange(N, device=dev(), dtype=torch.int) * 3
    rng = np.random.default_rng(0)
    buf.images[:N] = T(rng.random((N, 1, 3, ht * 8, wd * 8)), torch.float16)
    buf.masks[:N, 0, :4] = True
    buf.n_frames = N
    m = buf.extract_slam_map(filter_thresh=0.4)
    poses, disps = g.poses_gt.astype(np.float32), g.disps_gt.astype(np.float32)
    intr8 = (g.intrinsics[0] / 8.0).astype(np.float32)
    c2w = ose3.se3_inv(poses.astype(np.float64)).astype(np.float32)
    pts = frame_ops.iproj(c2w, disps, intr8)
    th = np.full(N, 0.4 / disps.mean(), np.float32)
    cnt = frame_ops.depth_filter(poses, disps, intr8, np.arange(N), th)
    mask = (cnt >= 2) & (disps > 0.5 * disps.mean(axis=(1, 2), keepdims=True))
    mask[:, :4] = False
    assert mask.sum() > 1000
    got_cnt = m.dense_disp_packinfo[:, 0, 1].cpu().numpy()
    assert np.abs(got_cnt - mask.reshape(N, -1).sum(1)).max() <= 2  # threshold ties
    if np.array_equal(got_cnt, mask.reshape(N, -1).sum(1)):
        assert np.allclose(m.dense_disp_xyz.cpu().numpy(), pts[mask], rtol=1e-4, atol=1e-4)
        rgb = buf.images[:N, 0][..., 3::8, 3::8].permute(0, 2, 3, 1).float().cpu().numpy()
        assert np.array_equal(m.dense_disp_rgb.float().cpu().numpy(), rgb[mask])
    assert m.dense_disp_frame_inds == [0, 3, 6, 9, 12, 15]
    xyz2, _ = m.get_dense_disp_pcd(2)
    assert xyz2.shape[0] == int(got_cnt[2])
    assert m.get_dense_disp_full_pcd()[0].shape[0] == int(got_cnt.sum())
    # project keyframe 2's own points (tstamp_nn = 0 window) back into keyframe 2 at 1/8 resolution
    depth = m.project_map(6, 0, (ht, wd), torch.from_numpy(intr8).to(dev()), SE3(buf.poses[2:3]).inv()[0], tstamp_nn=0)
    assert depth.shape == (ht, wd)
    # local map: points in the camera frame have z = 1 / disparity
    ml = buf.extract_slam_map(filter_thresh=0.4, is_local=True)
    z = ml.get_dense_disp_pcd(2)[0][:, 2].cpu().numpy()
    assert np.allclose(z, (1.0 / disps[2])[mask[2]], rtol=1e-5) or z.shape[0] != mask[2].sum()
    hit = depth > 0
    assert hit.float().mean() > 0.05
    ref_depth = torch.from_numpy(1.0 / disps[2]).to(dev())
    # a pixel-centre point lands on the pixel's edge (u = x exactly); compare with the nearest of the 2x2 neighbourhood
    err = torch.stack([(depth - torch.roll(ref_depth, (dy, dx), (0, 1))).abs() for dy in (0, -1) for dx in (0, -1)]).min(0).values
    assert float((err[hit] / ref_depth[hit]).median()) < 1e-3


@pytest.mark.parametrize("n_poses,intr,expect_lds_dense", [(4, False, None), (9, True, None), (17, False, True), (17, True, True),
                                                          (24, False, True), (27, False, True), (27, True, True), (28, False, False)])
def test_dense_window_solver_sizes(n_poses, intr, expect_lds_dense):
    """Sizes of the register-tile dense solver (`ba_solve_dense_kernel`: chain wave + fp64 matrix-core tiles, 6-column
    block steps, 16 x 16 tiles): windows with every pair coupled from one tile row to the kernel's limit of 160 rows
    (26 free poses; with the focal length n + 1 = 158), tile-row boundaries (n + 1 = 97: seven tile rows), the partial
    last block of the intrinsics column - and the first size beyond the limit, which another solver must take.
    HIP vs the fp64 oracle."""
    g = make_graph(n=n_poses, height=96, width=128, radius=n_poses - 1, seed=17 + n_poses)
    bk = dict(t0=1, t1=n_poses, n_iters=2, pose_damping=1e-3, pose_ep=0.1, motion_only=False, limited_disp=False,
              optimize_intrinsics=intr)
    p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole", bk)
    E = len(g.ii)
    op, od, ok_, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                           ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                           g.eta[:, None], g.ii, g.jj, **bk)
    assert info[0] == n_poses - 1 and info[3] == 6 * (n_poses - 1) + int(intr) and info[2] == 0
    if expect_lds_dense is not None:
        assert (info[5] == 2) == expect_lds_dense
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()
    if intr:
        assert np.abs(k - ok_).max() <= 1e-4 * np.abs(ok_).max()


def test_dense_window_solver_with_two_intrinsics_columns():
    """The register-tile dense solver with the unified (MEI) camera: focal length + distortion = TWO tail columns, i.e. a
    last block step of width 2 (and a 2-wide preview of it one step earlier).  17 poses with every pair coupled
    (n = 98: the tail starts a new 16 x 16 tile row).  HIP vs the fp64 oracle."""
    g = make_graph(n=17, height=96, width=128, radius=16, seed=53)
    intr = np.concatenate([g.intrinsics, np.array([[0.4]], np.float32)], 1)
    bk = dict(t0=1, t1=17, n_iters=2, pose_damping=1e-3, pose_ep=0.1, motion_only=False, limited_disp=False,
              optimize_intrinsics=True)
    p, d, k, info = run_hip_ba(g, intr, "mei", bk)
    E = len(g.ii)
    op, od, ok_, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], intr, ose3.se3_identity(1),
                                           g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2), g.eta[:, None], g.ii, g.jj,
                                           model="mei", **bk)
    assert info[0] == 16 and info[3] == 98 and info[2] == 0 and info[5] == 2
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()
    assert np.abs(k - ok_).max() <= 1e-4 * np.abs(ok_).max()


@pytest.mark.parametrize("intr", [False, True])
def test_dense_ba_dense_window_of_twelve_poses(intr):
    """The frontend's steady state (12 free poses, every pair coupled, source degree 12): the reduced system is a dense
    72 x 72 block matrix (+ focal) - the widest band the LDS solver takes (two band columns per lane in the back
    substitution, six trailing-update pairs per thread).  HIP fp32 vs the fp64 oracle."""
    g = make_graph(n=13, height=96, width=128, radius=12, seed=31)
    assert len(g.ii) == 156
    bk = dict(t0=1, t1=13, n_iters=2, pose_damping=1e-4, pose_ep=1e-2, motion_only=False, limited_disp=False,
              optimize_intrinsics=intr)
    p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole", bk)
    E = len(g.ii)
    op, od, ok_, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                           ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                           g.eta[:, None], g.ii, g.jj, **bk)
    assert info[0] == 12 and info[2] == 0
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()
    if intr:
        assert np.abs(k - ok_).max() <= 1e-4 * np.abs(ok_).max()


def test_corr_pool_matches_corr_block_through_add_and_remove():
    """CorrPool (pooled pyramids + slot indirection in the fused lookup kernel) against plain CorrBlock cat / index:
    identical lookups after appends, removals (slot reuse) and growth beyond the initial capacity."""
    from vipe_amd.ext import droid_net_ext
    from vipe_amd.slam.networks import CorrBlock, CorrPool
    from vipe_amd.slam.networks import UpdateModule
    torch.manual_seed(0)
    eng = UpdateModule().eval().engine(dev())
    g = torch.Generator().manual_seed(8)
    h, w = 8, 64

    def block(n):
        f1 = torch.randn(1, n, 128, h, w, generator=g).half().to(dev())
        f2 = torch.randn(1, n, 128, h, w, generator=g).half().to(dev())
        return CorrBlock(f1, f2)

    pool = CorrPool(capacity=4)
    ref = None
    steps = [("add", 3), ("rm", [0, 2]), ("add", 2), ("add", 4), ("rm", [1, 2, 3, 5]), ("add", 3)]
    for op, arg in steps:
        if op == "add":
            b = block(arg)
            lv = [l.clone() for l in b.corr_pyramid]
            pool.cat(b)
            ref = lv if ref is None else [torch.cat([r, l], 0) for r, l in zip(ref, lv)]
        else:
            keep = np.array(arg)
            pool = pool[keep]
            ref = [r[torch.from_numpy(keep).to(dev())] for r in ref]
        E = ref[0].shape[0]
        assert len(pool) == E and tuple(pool.slots.shape) == (E,)
        for a, b_ in zip(pool.corr_pyramid, ref):
            assert torch.equal(a, b_)
        coords = (torch.rand(E, h, w, 2, generator=g) * torch.tensor([w - 1.0, h - 1.0])).to(dev())
        handle = pool.lookup_deferred(coords)
        assert handle[0] == "lookup" and len(handle) == 5
        out_p = torch.zeros(E, h, w, 128, dtype=torch.float16, device=dev())
        out_r = torch.zeros_like(out_p)
        droid_net_ext.corr_lookup_conv1x1(handle[1], handle[2], eng.corr0.packed, eng.corr0.bias, out_p, act="relu",
                                          slots=handle[3])
        droid_net_ext.corr_lookup_conv1x1(ref, coords, eng.corr0.packed, eng.corr0.bias, out_r, act="relu")
        assert torch.equal(out_p, out_r)
        assert torch.equal(pool.lookup_nhwc(coords), droid_net_ext.corr_pyramid_lookup_nhwc(ref, coords, 3, 200))


def test_blocked_pool_built_from_frame_indices_matches_reference_layout():
    """The pooled store the update iteration reads: `CorrPool.add_edges` = `vipe_corr_pyramid_build_indexed` writing the
    BLOCKED layout straight into free slots from (frame i, frame j) index vectors.  Against `CorrBlock(fmaps[i], fmaps[j])`
    (gathered maps, reference layout): every level BIT-IDENTICAL after conversion (same MFMA accumulation and pooling
    arithmetic, only the addresses differ), the fused lookup + 1x1 conv kernel BIT-IDENTICAL on both layouts, through
    additions, removals with slot reuse, pool growth, and a reference-layout block appended with `cat`."""
    from vipe_amd.ext import droid_net_ext
    from vipe_amd.slam.networks import CorrBlock, CorrPool
    from vipe_amd.slam.networks import UpdateModule
    torch.manual_seed(0)
    eng = UpdateModule().eval().engine(dev())
    g = torch.Generator().manual_seed(21)
    for (h, w, nf) in ((8, 64, 7), (48, 64, 5), (16, 128, 4)):
        fmaps = torch.randn(nf, 128, h, w, generator=g).half().to(dev())
        pool = CorrPool(capacity=4)
        ii = torch.zeros(0, dtype=torch.long)
        jj = torch.zeros(0, dtype=torch.long)
        steps = [("add", 3), ("rm", [0, 2]), ("add", 4), ("cat", 2), ("rm", [1, 2, 4, 5, 6]), ("add", 3)]
        for op, arg in steps:
            if op in ("add", "cat"):
                i_new = torch.randint(0, nf, (arg,), generator=g)
                j_new = torch.randint(0, nf, (arg,), generator=g)
                if op == "add":
                    pool.add_edges(fmaps, i_new.to(dev()), j_new.to(dev()))
                else:
                    pool.cat(CorrBlock(fmaps[i_new.to(dev())][None], fmaps[j_new.to(dev())][None]))
                ii, jj = torch.cat([ii, i_new]), torch.cat([jj, j_new])
            else:
                keep = np.array(arg)
                pool = pool[keep]
                ii, jj = ii[keep], jj[keep]
            E = ii.shape[0]
            assert pool.blocked and len(pool) == E and pool.pool[0].dim() == 7
            ref = CorrBlock(fmaps[ii.to(dev())][None], fmaps[jj.to(dev())][None])
            for a, b_ in zip(pool.corr_pyramid, ref.corr_pyramid):
                assert a.shape == b_.shape and torch.equal(a, b_)
            coords = (torch.rand(E, h, w, 2, generator=g) * torch.tensor([w + 6.0, h + 6.0]) - 3.0).to(dev())
            hd = pool.lookup_deferred(coords)
            out_p = torch.zeros(E, h, w, 128, dtype=torch.float16, device=dev())
            out_r = torch.zeros_like(out_p)
            droid_net_ext.corr_lookup_conv1x1(hd[1], hd[2], eng.corr0.packed, eng.corr0.bias, out_p, act="relu", slots=hd[3])
            droid_net_ext.corr_lookup_conv1x1(ref.levels, coords, eng.corr0.packed, eng.corr0.bias, out_r, act="relu")
            assert torch.equal(out_p, out_r)
        # CorrBlock.from_buffer (the backend's per-chunk volume): blocked, no slots
        blk = CorrBlock.from_buffer(fmaps, ii.to(dev()), jj.to(dev()))
        assert blk.levels[0].dim() == 7
        hd = blk.lookup_deferred(coords)
        out_b = torch.zeros_like(out_p)
        droid_net_ext.corr_lookup_conv1x1(hd[1], hd[2], eng.corr0.packed, eng.corr0.bias, out_b, act="relu")
        assert torch.equal(out_b, out_r)
        assert torch.equal(blk(coords[None]), ref(coords[None]))  # reference-API lookup through the converter


def test_update_batch_merged_chunks_equal_reference_groups_of_eight(monkeypatch):
    """The reference applies the operator per group of 8 source keyframes (factor_graph.py:337-343); merging the groups
    into one chunk (the default here) must not change anything: GraphAgg only couples edges of the same source frame.
    20 keyframes -> 3 groups; VIPE_AMD_BACKEND_CHUNK_EDGES=1 forces one group per chunk like the reference."""
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.factor_graph import FactorGraph
    from vipe_amd.slam.networks import UpdateModule
    N = 20

    def run(chunk_edges, volume_gb=None):
        if chunk_edges is None:
            monkeypatch.delenv("VIPE_AMD_BACKEND_CHUNK_EDGES", raising=False)
        else:
            monkeypatch.setenv("VIPE_AMD_BACKEND_CHUNK_EDGES", str(chunk_edges))
        if volume_gb is None:
            monkeypatch.delenv("VIPE_AMD_BACKEND_VOLUME_GB", raising=False)
        else:
            monkeypatch.setenv("VIPE_AMD_BACKEND_VOLUME_GB", str(volume_gb))
        g = make_graph(n=N, height=64, width=512, radius=2, seed=53)
        buf = GraphBuffer(64, 512, buffer_size=N + 2, device=dev())
        buf.n_frames = N
        buf.poses[:N], buf.disps[:N, 0], buf.intrinsics[:] = T(g.poses), T(g.disps), T(g.intrinsics)
        gen = torch.Generator().manual_seed(53)
        buf.fmaps[:N, 0] = torch.randn(N, 128, 8, 64, generator=gen).half().to(dev())
        buf.nets[:N, 0] = torch.randn(N, 128, 8, 64, generator=gen).tanh().half().to(dev())
        buf.inps[:N, 0] = torch.randn(N, 128, 8, 64, generator=gen).relu().half().to(dev())
        buf.masks[3, 0, :2] = True
        torch.manual_seed(0)
        graph = FactorGraph(UpdateModule().eval(), buf, dev(), max_factors=-1, incremental=False)
        perm = np.random.default_rng(1).permutation(len(g.ii))  # edges not sorted by source frame
        graph.add_factors(torch.from_numpy(g.ii[perm]), torch.from_numpy(g.jj[perm]))
        graph.update_batch(itrs=2, steps=2, optimize_intrinsics=False, optimize_rig_rotation=False)
        torch.cuda.synchronize()
        return (graph.target.cpu().numpy(), graph.weight.cpu().numpy(), graph.net_n.float().cpu().numpy(),
                graph.damping[:N].cpu().numpy(), buf.poses[:N].cpu().numpy(), buf.disps[:N, 0].cpu().numpy())

    from vipe_amd.slam import factor_graph as fgm
    merged = run(None)
    grouped = run(1)
    # pyramids that do not fit the volume budget: as many chunks as fit next to the working one stay resident over the
    # passes, the others are rebuilt every pass - the same result, and the build count shows the partial residency
    E = len(make_graph(n=N, height=64, width=512, radius=2, seed=53).ii)
    built0 = fgm.WORK["pyramids_built"]
    partial = run(1, volume_gb=0.04)  # ~0.7 MB per edge on the 8 x 64 grid: room for ~60 of the 74 pyramids
    built = fgm.WORK["pyramids_built"] - built0
    assert E < built < 2 * E, (E, built)
    for a, b, tol in zip(merged, partial, (2e-3, 2e-3, 2e-3, 1e-4, 1e-4, 1e-4)):
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max()), (np.abs(a - b).max(), tol)
    # target / weight / hidden state / eta: the same kernels on the same rows; only the global-context mean inside a
    # workgroup reduction and the BA's atomics may reorder float sums (fp16 state: ~1e-3 px on the targets, 1e-5 on the map)
    for a, b, tol in zip(merged, grouped, (2e-3, 2e-3, 2e-3, 1e-4, 1e-4, 1e-4)):
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max()), (np.abs(a - b).max(), tol)


def test_backend_global_ba_pass():
    """SLAMBackend.run (backend.py:73-117): fresh non-incremental graph with max_factors = 16 t, proximity edges with the
    backend thresholds (the neighbourhood part is deterministic: 3 predecessors per keyframe, both directions), `steps`
    passes of update_batch.  Perfect-geometry buffer: the global BA must keep poses / disparities finite and close."""
    from vipe_amd.slam.backend import BackendArgs, SLAMBackend
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.networks import UpdateModule
    N = 10
    g = make_graph(n=N, height=64, width=512, radius=2, seed=61)
    buf = GraphBuffer(64, 512, buffer_size=N + 2, device=dev())
    buf.n_frames = N
    buf.poses[:N], buf.disps[:N, 0], buf.intrinsics[:] = T(g.poses_gt, torch.float32), T(g.disps_gt, torch.float32), T(g.intrinsics)
    gen = torch.Generator().manual_seed(61)
    buf.fmaps[:N, 0] = torch.randn(N, 128, 8, 64, generator=gen).half().to(dev())
    buf.nets[:N, 0] = torch.randn(N, 128, 8, 64, generator=gen).tanh().half().to(dev())
    buf.inps[:N, 0] = torch.randn(N, 128, 8, 64, generator=gen).relu().half().to(dev())
    torch.manual_seed(0)
    be = SLAMBackend(UpdateModule().eval(), buf, BackendArgs(), dev())
    graph = be.run(steps=2)
    ii, jj = graph.ii.cpu().numpy(), graph.jj.cpu().numpy()
    assert graph.max_factors == 16 * N and not graph.incremental
    es = set(zip(ii.tolist(), jj.tolist()))
    for i in range(N):
        for j in range(max(i - 3, 0), i):
            assert (i, j) in es and (j, i) in es
    assert len(es) == len(ii)  # no repeated edge
    assert tuple(graph.target.shape) == (1, len(ii), 8, 64, 2)
    assert bool(torch.isfinite(buf.poses[:N]).all()) and bool(torch.isfinite(buf.disps[:N]).all())
    assert float(buf.disps[:N].min()) >= 1e-3
    assert torch.equal(buf.poses[0].cpu(), torch.from_numpy(g.poses_gt[0]).float())  # pose 0 is the gauge
    # a single keyframe: the graph stays empty and the sensor depth is taken where present
    buf1 = GraphBuffer(64, 512, buffer_size=2, device=dev())
    buf1.n_frames = 1
    buf1.disps_sens[0, 0, :4] = 0.5
    g1 = SLAMBackend(UpdateModule().eval(), buf1, BackendArgs(), dev()).run(steps=1)
    assert len(g1.ii) == 0 and float(buf1.disps[0, 0, :4].max()) == 0.5 and float(buf1.disps[0, 0, 4:].min()) == 1.0


@pytest.mark.parametrize("intr", [False, True])
def test_dense_ba_long_trajectory_band_exceeds_lds(intr):
    """120 keyframes, radius-3 graph: the band of the 714-unknown reduced system (245 KB in fp64) exceeds the LDS, so the
    blocked global-memory Cholesky (which still restricts its trailing update to the band) takes it.  HIP fp32 vs the
    fp64 oracle."""
    g = make_graph(n=120, height=96, width=128, radius=3, seed=17)
    bk = dict(t0=1, t1=120, n_iters=2, pose_damping=1e-4, pose_ep=1e-2, motion_only=False, limited_disp=False,
              optimize_intrinsics=intr)
    p, d, k, info = run_hip_ba(g, g.intrinsics, "pinhole", bk)
    E = len(g.ii)
    op, od, ok_, _ = oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics,
                                           ose3.se3_identity(1), g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2),
                                           g.eta[:, None], g.ii, g.jj, **bk)
    assert info[0] == 119 and info[2] == 0
    assert np.abs(p - op).max() <= 1e-4 * max(1.0, np.abs(op).max())
    assert np.abs(d - od[:, 0]).max() <= 1e-4 * np.abs(od).max()
    if intr:
        assert np.abs(k - ok_).max() <= 1e-4 * np.abs(ok_).max()


def test_ba_plan_reuse_equals_rebuilding_the_plan(monkeypatch):
    """FactorGraph keeps a private BA workspace and tells the library when the edge plan in it is still the one of the
    previous call (vipe_ba_params.reuse_plan): three update iterations with reuse must equal three with the plan rebuilt
    every call - including when the sensor depth of a frame appears between two calls (data the plan must not freeze)."""
    import bench

    from vipe_amd.ext import slam_ext

    def run(no_reuse):
        monkeypatch.setattr(slam_ext, "PLAN_REUSE", not no_reuse)
        g, buf, graph = bench.build_problem(dev(), 12, 384, 512, 3, 0, seed=7, depth_prior=False)
        out = []
        for it in range(3):
            if it == 2:
                buf.disps_sens[5, 0] = buf.disps[5, 0] * 1.05  # frame 5 gets sensor depth before the third call
            graph.update(t0=1, t1=12, itrs=2)
            torch.cuda.synchronize()
            out.append((buf.poses[:12].cpu().numpy().copy(), buf.disps[:12, 0].cpu().numpy().copy()))
        return out, graph

    a, ga = run(False)
    b, _ = run(True)
    assert ga._ba_state.get("key") is not None and ga._ba_state["ws"].numel() > 0
    for (pa, da), (pb, db) in zip(a, b):
        # two separate runs differ by the order of float atomics through three fp16 operator applications
        assert np.abs(pa - pb).max() <= 1e-4 * max(1.0, np.abs(pb).max())
        assert np.abs(da - db).max() <= 1e-4 * np.abs(db).max()
    # the third call saw the new sensor depth: frame 5 moved differently than it would have without it
    assert np.abs(a[2][1][5] - a[1][1][5]).max() > 0


def test_slam_system_two_passes_over_rgb_frames():
    """`SLAMSystem.run` (system.py:186-316) on a short synthetic clip: pass 1 (motion filter -> keyframes -> frontend,
    the backend at the configured keyframe counts, the two final global-BA passes), pass 2 (every frame through the
    InnerFiller), the map.  Random-init weights: what is pinned is the bookkeeping - which frames became keyframes, one
    pose per input frame, unit quaternions, the keyframe timestamps of the map - with the filter scripted to keep
    every third frame in the second case (so the context encoder's no-reuse branch and real interpolation run)."""
    from vipe_amd.ext.lietorch import SE3
    from vipe_amd.slam.frontend import FrontendArgs
    from vipe_amd.slam.inner_filler import InfillArgs
    from vipe_amd.slam.system import Frame, SLAMConfig, SLAMSystem

    gen = torch.Generator().manual_seed(5)
    T, H, W = 26, 128, 512
    rgb = torch.rand(T, H, W, 3, generator=gen).to(dev())
    depth = (1.0 + 4.0 * torch.rand(T, H, W, generator=gen)).to(dev())
    intr = torch.tensor([460.8, 460.8, 256.0, 64.0])

    def frames():
        out = []
        for t in range(T):
            pose = SE3(torch.tensor([[-0.05 * t, 0, 0, 0, 0, 0, 1.0]], device=dev())).inv()  # camera -> world
            out.append(Frame(rgb=rgb[t], metric_depth=depth[t], intrinsics=intr, pose=SE3(pose.data[0]),
                             mask=torch.ones(H, W, dtype=torch.bool, device=dev())))
        return out

    torch.manual_seed(0)
    for every in (1, 3):
        cfg = SLAMConfig(buffer=64, filter_thresh=0.0, frontend_backend_iters=(10,),
                         frontend=FrontendArgs(keyframe_thresh=0.0), infill=InfillArgs(infill_chunk_size=8))
        sysm = SLAMSystem(dev(), cfg)
        calls = {"backend": 0}
        if every > 1:
            import vipe_amd.slam.system as S
            real_build = sysm._build_components

            def build(*a, **k):
                real_build(*a, **k)
                mf, state = sysm.motion_filter, {"i": -1}
                real_check = mf.check

                def check(images, masks=None):
                    state["i"] += 1
                    return real_check(images, masks) if state["i"] % every == 0 else False
                mf.check = check
                real_run = sysm.backend.run_if_necessary

                def run_if(*a, **k):
                    calls["backend"] += 1
                    return real_run(*a, **k)
                sysm.backend.run_if_necessary = run_if
            sysm._build_components = build
        out = sysm.run(frames())
        torch.cuda.synchronize()
        want = sorted(set(range(0, T, every)) | {T - 1})
        assert out.keyframe_ids.tolist() == want
        assert out.trajectory.data.shape == (T, 7) and bool(torch.isfinite(out.trajectory.data).all())
        assert (out.trajectory.data[:, 3:].norm(dim=-1) - 1).abs().max().item() < 1e-4
        assert out.intrinsics.shape == (1, 4) and torch.allclose(out.intrinsics[0].cpu(), intr)
        assert out.get_view_trajectory(0).data.shape == (T, 7)
        assert sysm.buffer.n_frames == len(want) and out.slam_map is not None
        assert bool((sysm.buffer.masks[:len(want)] == 0).all())  # all-valid masks -> nothing marked invalid
        if every > 1:
            assert calls["backend"] == 1  # 10 keyframes reached once


@pytest.mark.parametrize("camera", ["pinhole", "mei"])
def test_fused_frame_distance_equals_the_operator_sequence(camera):
    """`GraphBuffer.frame_distance_dense_disp` in one launch (`vipe_frame_distance_rig`: per-view poses R_v^-1 G_n,
    1/8-scale pinhole intrinsics, both directions, the mean) against the reference's sequence of operators it replaces
    (lietorch inv / mul over all frames, two `frame_distance` calls, the average: `fused=False`): the SAME bits - the edge
    proposal compares these numbers with thresholds and orders by them.  One camera and a two-camera rig with a view
    offset, pinhole and MEI, one- and two-directional."""
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.ext.lietorch import SE3
    gen = torch.Generator().manual_seed(9)
    for V in (1, 2):
        n = 7
        buf = GraphBuffer(96, 128, n_views=V, buffer_size=8, camera_type=camera, device=dev())
        buf.n_frames = n
        xi = torch.randn(8, 6, generator=gen) * torch.tensor([0.2, 0.2, 0.2, 0.05, 0.05, 0.05])
        buf.poses[:] = SE3.exp(xi.to(dev())).data
        buf.disps[:] = (0.2 + torch.rand(8, V, 12, 16, generator=gen)).to(dev())
        k = [110.0, 105.0, 64.0, 48.0] + ([0.3] if camera == "mei" else [])
        buf.intrinsics[:] = torch.tensor([k, [x * 1.03 for x in k]][:V], device=dev())
        if V == 2:
            buf.rig[1] = SE3.exp(torch.tensor([[0.3, 0.02, -0.01, 0.01, 0.2, -0.02]], device=dev())).data[0]
        ii, jj = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
        ii, jj = ii.reshape(-1).to(dev()), jj.reshape(-1).to(dev())
        for bidir in (True, False):
            for off in range(V):
                a = buf.frame_distance_dense_disp(ii, jj, beta=0.3, bidirectional=bidir, view_offset=off)
                b = buf.frame_distance_dense_disp(ii, jj, beta=0.3, bidirectional=bidir, view_offset=off, fused=False)
                assert a.shape == b.shape == (n * n, V) and torch.equal(a, b), (V, bidir, off, (a - b).abs().max().item())
        assert float(a.max()) > 1.0  # not vacuous


def test_frontend_next_frame_kernel_matches_the_group_operations():
    """`vipe_frontend_next_frame` (frontend.py:70-76 + :118-122 / :147-151 in one launch) against the same steps through
    the lietorch operators and torch means: constant-velocity pose, per-view mean disparity over the last 1 / 4 keyframes,
    poses untouched when the caller supplies them."""
    from vipe_amd._lib import check, lib, ptr, stream_ptr
    from vipe_amd.ext.lietorch import SE3
    gen = torch.Generator().manual_seed(3)
    N, V, h, w = 9, 2, 5, 7
    xi = torch.randn(N, 6, generator=gen) * torch.tensor([0.3, 0.3, 0.3, 0.2, 0.2, 0.2])
    poses0 = SE3.exp(xi.to(dev())).data.contiguous()
    disps0 = torch.rand(N, V, h, w, generator=gen).to(dev()) + 0.1
    for t1, n_mean, init in ((6, 1, 1), (8, 4, 1), (5, 1, 0)):
        poses, disps = poses0.clone(), disps0.clone()
        check(lib().vipe_frontend_next_frame(ptr(poses), ptr(disps), t1, V, h * w, n_mean, init, stream_ptr(poses)), "next_frame")
        want_p = poses0.clone()
        if init:
            p1, p2 = SE3(poses0[t1 - 2][None]), SE3(poses0[t1 - 1][None])
            want_p[t1] = (SE3.exp((p2 * p1.inv()).log() * 0.5) * p2).data[0]
        assert torch.allclose(poses, want_p, atol=1e-6) and torch.equal(poses[:t1], poses0[:t1])
        want_d = disps0.clone()
        for v in range(V):
            want_d[t1, v] = disps0[t1 - n_mean:t1, v].mean()
        assert torch.allclose(disps, want_d, rtol=1e-6, atol=1e-7)


def test_edge_stores_append_and_compact_match_cat_and_index():
    """`factor_graph._EdgeStores` (hidden state, operator input, gate context of an incremental graph in banked buffers
    with spare capacity): `append` = the reference's gather + permute + cat of `nets[ii]` / `inps[ii]`
    (factor_graph.py:147-170) in one launch, `compact` = its boolean-mask compaction (:190-201) of every tensor - and of
    extra small arrays - in one launch; growth beyond the capacity and a hidden state assigned from outside included."""
    from vipe_amd.slam.factor_graph import _EdgeStores
    h, w, N = 5, 9, 7
    gen = torch.Generator().manual_seed(0)
    nets = torch.randn(N, 128, h, w, generator=gen).half().to(dev())
    inps = torch.randn(N, 128, h, w, generator=gen).half().to(dev())
    st = _EdgeStores(h, w, dev(), with_pgate=True)
    net_n = xbuf = None
    ref_net = torch.zeros(0, h, w, 128, dtype=torch.float16, device=dev())
    ref_inp = ref_net.clone()
    ref_pg = torch.zeros(0, h, w, 384, dtype=torch.float16, device=dev())
    small = torch.zeros(1, 0, h, w, 2, device=dev())
    for step in range(14):
        k = int(torch.randint(1, 40 if step == 9 else 9, (1,), generator=gen))  # step 9 outgrows the 64-row capacity
        fr = torch.randint(0, N, (k,), generator=gen).to(dev())
        n0 = ref_net.shape[0]
        net_n, xbuf, xb_new = st.append(nets.view(N, 128, -1), inps.view(N, 128, -1), fr, net_n, n0)
        pg_new = torch.randn(k, h, w, 384, generator=gen).half().to(dev())
        st.pg[st.cur][n0:n0 + k] = pg_new
        ref_net = torch.cat([ref_net, nets[fr].permute(0, 2, 3, 1)], 0)
        ref_inp = torch.cat([ref_inp, inps[fr].permute(0, 2, 3, 1)], 0)
        ref_pg = torch.cat([ref_pg, pg_new], 0)
        small = torch.cat([small, torch.randn(1, k, h, w, 2, generator=gen).to(dev())], 1)
        assert torch.equal(net_n, ref_net) and torch.equal(xbuf[..., :128], ref_inp) and torch.equal(xb_new, xbuf[n0:])
        assert torch.equal(st.pg[st.cur][:n0 + k], ref_pg)
        if step % 4 == 1:  # the operator ping-pongs the hidden state into the other bank
            sp = st.net_spare(net_n)
            assert sp.shape == net_n.shape and sp.data_ptr() != net_n.data_ptr()
            sp.copy_(net_n * 0.5)
            net_n, ref_net = sp, ref_net * 0.5
        if step == 6:  # a state assigned from outside (tests restore snapshots): adopted on the next append
            net_n = net_n.clone()
        if step % 3 == 2:
            keep = torch.nonzero(torch.rand(ref_net.shape[0], generator=gen) > 0.3).reshape(-1).to(dev())
            nk = int(keep.shape[0])
            if st.bank_of(net_n) is None:
                net_n = st.reserve(net_n.shape[0], net_n, net_n.shape[0])
            out_small = torch.empty((1, nk, h, w, 2), device=dev())
            net_n, xbuf, pg = st.compact(keep, nk, net_n, [(small.contiguous(), out_small, keep, nk, h * w * 2 * 4, 0)])
            ref_net, ref_inp, ref_pg, small = ref_net[keep], ref_inp[keep], ref_pg[keep], small[:, keep]
            assert torch.equal(net_n, ref_net) and torch.equal(xbuf[..., :128], ref_inp) and torch.equal(pg, ref_pg)
            assert torch.equal(out_small, small)
            small = out_small


def test_operator_api_on_empty_inputs():
    """Zero-sized batches through every operator of the boundary: an empty edge list / point set / element batch returns
    correctly shaped empty (or untouched) outputs and launches nothing - a zero-block grid is a launch error in HIP, and a
    pipeline that proposes no edge for a keyframe or filters every point of a map must not die on it."""
    from vipe_amd.ext import corr_ext, droid_net_ext, lietorch_ext, scatter_ext, slam_ext, utils_ext
    d = dev()
    f32 = dict(device=d, dtype=torch.float32)
    i64 = dict(device=d, dtype=torch.int64)
    e = torch.zeros(0, **i64)
    poses = torch.zeros(4, 7, **f32)
    poses[:, 6] = 1
    disps = torch.ones(4, 12, 16, **f32)
    intr = torch.tensor([50.0, 50.0, 8.0, 6.0], **f32)
    # slam_ext
    assert slam_ext.frame_distance(poses, disps, intr[None], e, e, e, e, e, 0.3).shape == (0,)
    coords, valid = slam_ext.projmap(poses, disps, intr, e, e)
    assert coords.shape == (0, 12, 16, 3) and valid.shape == (0, 12, 16, 1)
    assert slam_ext.depth_filter(poses[:0], disps[:0], intr, e, torch.zeros(0, **f32)).shape == (0, 12, 16)
    assert slam_ext.iproj(poses[:0], disps[:0], intr).shape == (0, 12, 16, 3)
    rig = torch.zeros(1, 7, **f32)
    rig[:, 6] = 1
    c, v = slam_ext.reproject(poses, disps, intr[None], rig, e, e, e, e, e)
    assert c.shape[0] == 0 and v.shape[0] == 0
    # droid_net_ext
    vol = torch.zeros(0, 12, 16, 12, 16, device=d, dtype=torch.float16)
    (o,) = droid_net_ext.corr_index_forward(vol, torch.zeros(0, 2, 12, 16, **f32), 3)
    assert o.shape == (0, 7, 7, 12, 16)
    (g,) = droid_net_ext.corr_index_backward(vol.float(), torch.zeros(0, 2, 12, 16, **f32), torch.zeros(0, 7, 7, 12, 16, **f32), 3)
    assert g.shape == vol.shape
    f1 = torch.zeros(1, 12, 16, 32, **f32)
    (o,) = droid_net_ext.altcorr_forward(f1, f1, torch.zeros(1, 0, 12, 16, 2, **f32), 3)
    assert o.shape == (1, 0, 49, 12, 16)
    # lietorch_ext (gid 3 = SE3)
    X0, a0 = torch.zeros(0, 7, **f32), torch.zeros(0, 6, **f32)
    assert lietorch_ext.expm(3, a0).shape == (0, 7) and lietorch_ext.logm(3, X0).shape == (0, 6)
    assert lietorch_ext.inv(3, X0).shape == (0, 7) and lietorch_ext.mul(3, X0, X0).shape == (0, 7)
    assert lietorch_ext.adjT(3, X0, a0).shape == (0, 6) and lietorch_ext.act4(3, X0, torch.zeros(0, 4, **f32)).shape == (0, 4)
    assert lietorch_ext.as_matrix(3, X0).shape == (0, 4, 4)
    # scatter_ext
    out = scatter_ext.scatter_sum(torch.zeros(0, 5, **f32), e, 0, None, 3)
    assert out.shape == (3, 5) and float(out.abs().sum()) == 0
    out = scatter_ext.scatter_mean(torch.zeros(0, 5, **f32), e, 0, None, 3)
    assert out.shape == (3, 5) and bool(torch.isfinite(out).all())
    mx, arg = scatter_ext.scatter_max(torch.zeros(0, 5, **f32), e, 0, None, 3)
    assert mx.shape == (3, 5) and arg.shape == (3, 5)
    # corr_ext, utils_ext
    x = torch.zeros(0, 8, 6, 6, **f32)
    assert corr_ext.forward(x, x, 1, 1, 3, 3, 0, 0, 1, 1, 1, 1, 1, 1).shape == (0, 3, 3, 6, 6)
    dist, idx = utils_ext.nearest_neighbours(torch.zeros(0, 3, **f32), torch.rand(10, 3, **f32), 2)
    assert dist.shape == (0, 2) and idx.shape == (0, 2)
    torch.cuda.synchronize()


def test_slam_system_on_a_two_camera_rig():
    """`SLAMSystem.run` with two views per frame and a fixed rig (system.py:221-231: per-view intrinsics from the first
    frame, `buffer.rig`, cross-view edges in frontend and backend): bookkeeping and finiteness of the whole two-pass
    run, and the per-view trajectories `trajectory * rig[v]` of the output."""
    from vipe_amd.ext.lietorch import SE3
    from vipe_amd.slam.frontend import FrontendArgs
    from vipe_amd.slam.inner_filler import InfillArgs
    from vipe_amd.slam.system import Frame, SLAMConfig, SLAMSystem

    gen = torch.Generator().manual_seed(9)
    T, V, H, W = 14, 2, 128, 512
    rgb = torch.rand(T, V, H, W, 3, generator=gen).to(dev())
    depth = (1.0 + 4.0 * torch.rand(T, V, H, W, generator=gen)).to(dev())
    intr = torch.tensor([[460.8, 460.8, 256.0, 64.0], [455.0, 455.0, 250.0, 66.0]])
    rig = SE3(torch.tensor([[0, 0, 0, 0, 0, 0, 1.0], [-0.1, 0, 0, 0, 0, 0, 1.0]], device=dev()))
    frames = []
    for t in range(T):
        pose = SE3(torch.tensor([-0.05 * t, 0, 0, 0, 0, 0, 1.0], device=dev())).inv()
        frames.append([Frame(rgb=rgb[t, v], metric_depth=depth[t, v], intrinsics=intr[v],
                             pose=(pose * rig[v]) if v else pose) for v in range(V)])
    torch.manual_seed(0)
    cfg = SLAMConfig(buffer=48, filter_thresh=0.0, frontend_backend_iters=(10,),
                     frontend=FrontendArgs(keyframe_thresh=0.0), infill=InfillArgs(infill_chunk_size=8))
    sysm = SLAMSystem(dev(), cfg)
    out = sysm.run(frames, rig=rig)
    torch.cuda.synchronize()
    assert out.keyframe_ids.tolist() == list(range(T)) and sysm.buffer.n_views == 2
    assert out.trajectory.data.shape == (T, 7) and bool(torch.isfinite(out.trajectory.data).all())
    assert (out.trajectory.data[:, 3:].norm(dim=-1) - 1).abs().max().item() < 1e-4
    assert torch.allclose(out.intrinsics.cpu(), intr) and torch.allclose(out.rig.data, rig.data)
    v1 = out.get_view_trajectory(1)
    assert v1.data.shape == (T, 7) and bool(torch.isfinite(v1.data).all())
    assert bool(torch.isfinite(sysm.buffer.disps[:T]).all()) and bool((sysm.buffer.disps[:T] >= 1e-3).all())


def test_spatial_correlation_sampler_module_and_autograd():
    """`vipe.ext.corr.SpatialCorrelationSampler` on device tensors: values and gradients through autograd vs an explicit
    torch formulation."""
    from vipe_amd.ext.corr import SpatialCorrelationSampler
    torch.manual_seed(4)
    a = torch.randn(2, 6, 9, 12, device=dev(), requires_grad=True)
    b = torch.randn(2, 6, 9, 12, device=dev(), requires_grad=True)
    patch, dil = (5, 3), (1, 2)
    out = SpatialCorrelationSampler(1, patch, 1, 0, 1, dil)(a, b)
    H, W = 9, 12
    rH, rW = dil[0] * (patch[0] - 1) // 2, dil[1] * (patch[1] - 1) // 2
    bp = torch.nn.functional.pad(b, (rW, rW, rH, rH))
    ref = torch.stack([torch.stack([(a * bp[:, :, ph * dil[0]:ph * dil[0] + H, pw * dil[1]:pw * dil[1] + W]).sum(1)
                                    for pw in range(patch[1])], 1) for ph in range(patch[0])], 1)
    assert out.shape == ref.shape and torch.allclose(out, ref, atol=1e-4)
    go = torch.randn_like(ref)
    g = torch.autograd.grad(out, (a, b), go)
    r = torch.autograd.grad(ref, (a, b), go)
    assert torch.allclose(g[0], r[0], atol=1e-4) and torch.allclose(g[1], r[1], atol=1e-4)


@pytest.mark.parametrize("name", ["n5_frontend", "n6_window_prior", "n6_infill_motion_limited"])
def test_dense_ba_with_a_sparse_track_term_matches_reference_solver(name):
    """The second flow term of buffer.py:422-447 (sparse tracks enabled) through `GraphBuffer.bundle_adjustment` with a
    caller-supplied tracker object: HIP BA on the folded term vs the reference Solver run with BOTH terms (fixture)."""
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.synth import make_tracks
    names = ["n5_frontend", "n6_window_prior", "n6_infill_motion_limited"]
    G = np.load(os.path.join(GOLD, "ba_tracks_reference.npz"))
    gk, bk = BA_CASES[name]
    g = make_graph(**gk)
    tt, tw = make_tracks(g, 100 + names.index(name))
    E, n = len(g.ii), g.poses.shape[0]

    class Tracker:
        enabled, calls = True, 0

        def compute_dense_disp_target_weight(self, source_view_inds, source_frame_inds, target_view_inds, target_frame_inds,
                                             image_size, dense_disp_size):
            assert image_size == (gk["height"], gk["width"]) and dense_disp_size == (g.ht, g.wd)
            assert source_frame_inds.cpu().tolist() == (10 * g.ii).tolist()  # time stamps, not buffer rows
            Tracker.calls += 1
            return T(tt.reshape(E, g.ht, g.wd, 2)), T(tw.reshape(E, g.ht, g.wd, 2))

    buf = GraphBuffer(gk["height"], gk["width"], buffer_size=n + 2, device=dev())
    buf.n_frames = n
    buf.poses[:n], buf.disps[:n, 0], buf.disps_sens[:n, 0] = T(g.poses), T(g.disps), T(g.disps_sens)
    buf.intrinsics[:] = T(g.intrinsics)
    buf.tstamp[:n] = 10 * torch.arange(n, device=dev(), dtype=torch.int)
    buf.sparse_tracks = Tracker()
    damping = torch.zeros(n + 2, g.ht, g.wd, device=dev())
    damping[:n] = T(g.eta)
    kw = dict(bk)
    buf.bundle_adjustment(T(g.target.reshape(E, -1, 2)), T(g.weight.reshape(E, -1, 2)), damping, T(g.ii), T(g.jj),
                          kw.pop("t0"), kw.pop("t1"), kw.pop("n_iters"), kw.pop("pose_damping"), kw.pop("pose_ep"),
                          kw.pop("motion_only"), kw.pop("limited_disp"), kw.pop("optimize_intrinsics"), False)
    torch.cuda.synchronize()
    p, d = buf.poses[:n].cpu().numpy(), buf.disps[:n, 0].cpu().numpy()
    rp, rd = G[name + "/poses"], G[name + "/disps"]
    assert Tracker.calls == 1
    assert np.abs(p - rp).max() <= 1e-4 * max(1.0, np.abs(rp).max())
    assert np.abs(d - rd).max() <= 1e-4 * np.abs(rd).max()


def test_slam_system_with_a_caller_supplied_tracker():
    """The tracker hooks of the two-pass driver: `track_image` once per frame of pass 1 (system.py:255), the track score in
    the motion filter (a tracker that loses all its keypoints forces keyframes), the track term in every BA."""
    from vipe_amd.slam.frontend import FrontendArgs
    from vipe_amd.slam.inner_filler import InfillArgs
    from vipe_amd.slam.system import Frame, SLAMConfig, SLAMSystem

    class Tracker:
        enabled = True

        def __init__(self):
            self.frames, self.ba_calls = 0, 0

        def track_image(self, frames):
            self.frames += 1

        def get_correspondences(self, view_idx, a, b):  # ten tracks on even frames, none on odd ones
            return torch.arange(10 if (a % 2 == 0 and b % 2 == 0) else 0)

        def get_observations(self, view_idx, frame, kp):
            return torch.zeros(len(kp), 2) + 0.01 * frame

        def compute_dense_disp_target_weight(self, source_view_inds, source_frame_inds, target_view_inds, target_frame_inds,
                                             image_size, dense_disp_size):
            self.ba_calls += 1
            E = source_view_inds.shape[0]
            z = torch.zeros(E, *dense_disp_size, 2, device=source_view_inds.device)
            return z, z.clone()  # weight 0 everywhere: the folded term equals the dense one

    gen = torch.Generator().manual_seed(6)
    T, H, W = 16, 128, 512
    rgb = torch.rand(T, H, W, 3, generator=gen).to(dev())
    intr = torch.tensor([460.8, 460.8, 256.0, 64.0])
    frames = [Frame(rgb=rgb[t], intrinsics=intr) for t in range(T)]
    torch.manual_seed(0)
    tr = Tracker()
    cfg = SLAMConfig(buffer=48, filter_thresh=0.0, frontend_backend_iters=(),
                     frontend=FrontendArgs(keyframe_thresh=0.0), infill=InfillArgs(infill_chunk_size=8))
    out = SLAMSystem(dev(), cfg, sparse_tracks=tr).run(frames)
    torch.cuda.synchronize()
    assert tr.frames == T and tr.ba_calls > 0
    assert out.trajectory.data.shape == (T, 7) and bool(torch.isfinite(out.trajectory.data).all())


def test_sparse_tracks_target_weight_matches_oracle():
    """`SparseTracks.compute_dense_disp_target_weight` on the device (the library's atomic scatter kernel) vs the numpy
    restatement, and through `GraphBuffer.bundle_adjustment` as the BA's second flow term."""
    from oracle import tracks
    from test_oracle_golden import _synthetic_tracks
    from vipe_amd.slam.sparse_tracks import ReplayedSparseTracks
    tr = ReplayedSparseTracks(_synthetic_tracks(9, n_frames=8, n_kp=200, size=(384, 512)))
    for _ in range(8):
        tr.track_image(None)
    ii = np.array([0, 1, 2, 3, 7, 4, 0, 6]); jj = np.array([1, 0, 4, 3, 2, 5, 7, 6])
    z = torch.zeros(len(ii), dtype=torch.long, device=dev())
    val, wgt = tr.compute_dense_disp_target_weight(z, T(ii), z, T(jj), (384, 512), (48, 64))
    rv, rw = tracks.dense_disp_target_weight(tr.observations, [0] * len(ii), ii.tolist(), jj.tolist(), (384, 512), (48, 64))
    assert np.abs(wgt.cpu().numpy() - rw).max() < 1e-5 and np.abs(val.cpu().numpy() - rv).max() < 1e-4
    assert (rw > 0).sum() > 500


@pytest.mark.parametrize("n,radius", [(48, 3), (33, 2), (34, 1), (40, 3)])
def test_two_chain_band_solve_equals_one_chain(n, radius, monkeypatch):
    """Long pose-only neighbourhood chains are eliminated from both ends at once (band2_solve_body: chain A in natural order,
    chain B mirrored, the separator's Schur contributions merged, separator factored, both chains back-substituted in
    parallel).  Same fp64 arithmetic on another elimination order: poses / disparities agree with the one-chain form
    (`solver_options=BA_OPT_ONE_CHAIN`) to 1e-5 (the north_star tolerance is 1e-4), for even and odd chain splits and several
    band widths."""
    bk = dict(t0=1, t1=n, n_iters=2, pose_damping=1e-3, pose_ep=0.1, motion_only=False, limited_disp=False,
              optimize_intrinsics=False)
    g = make_graph(n=n, height=96, width=128, radius=radius, seed=100 + n)
    res = {}
    for flag in ("1", "0"):
        res[flag] = run_hip_ba(g, g.intrinsics, "pinhole", dict(bk, solver_options=0 if flag == "1" else 1))
    (p1, d1, _, i1), (p0, d0, _, i0) = res["1"], res["0"]
    assert i1[2] == 0 and i0[2] == 0 and i1[5] == 1 and i0[5] == 1  # no failed pivot; the LDS band solver took both
    # (fp32 states; the accumulate kernels' atomics alone make two runs of ONE form differ in the last bits)
    assert np.abs(p1 - p0).max() <= 1e-5 * max(1.0, np.abs(p0).max()) and np.abs(d1 - d0).max() <= 1e-5 * np.abs(d0).max()
    assert np.abs(p0 - g.poses).max() > 1e-4  # the step moved the poses
