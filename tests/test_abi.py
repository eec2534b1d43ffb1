"""The C-ABI library loads without a GPU and exports every symbol include/vipe_amd.h declares."""

import ctypes
import os

import numpy as np
import torch

from vipe_amd import _lib
from oracle import se3 as ose3


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 40
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(raw, n)]
    assert not missing, missing
    L = _lib.lib()
    assert b"gfx950" in L.vipe_amd_version()
    assert L.vipe_amd_abi_version() == 4


def test_header_cites_reference_interfaces():
    src = open(_lib.HEADER).read()
    for cite in ["droid.cpp", "slam.cpp", "lietorch.cpp", "scatter.cpp", "correlation_sampler.cpp", "bind.cpp"]:
        assert cite in src


def test_argument_errors_do_not_launch():
    L = _lib.lib()
    assert L.vipe_corr_index_forward(None, None, None, 1, 4, 4, 4, 4, 3, 0, None) == -1
    assert L.vipe_corr_index_forward(None, None, None, 0, 4, 4, 4, 4, 3, 0, None) == 0  # empty batch
    p = _lib.BAParams(n_poses=0)
    assert L.vipe_dense_ba_workspace_bytes(ctypes.byref(p)) < 0
    p = _lib.BAParams(n_poses=48, n_views=1, ht=48, wd=64, M=276)
    nbytes = L.vipe_dense_ba_workspace_bytes(ctypes.byref(p))
    assert 20e6 < nbytes < 200e6  # E_j rows dominate: 276 * 6 * 3072 * 4 B


def test_lietorch_host_path_matches_oracle():
    """lietorch_ext on CPU tensors runs the library's host loop over the same closed forms as the kernels."""
    from vipe_amd.ext.lietorch import SE3, SO3
    rng = np.random.default_rng(0)
    xi = rng.normal(0, 1, (200, 6))
    xi[:, 3:] *= np.minimum(1.0, 3.0 / np.linalg.norm(xi[:, 3:], axis=-1, keepdims=True))  # |phi| < pi
    xi[:5, 3:] *= 1e-9
    X = SE3.exp(torch.from_numpy(xi))
    Xo = ose3.se3_exp(xi)
    assert np.abs(X.data.numpy() - Xo).max() < 1e-12
    assert np.abs(X.log().numpy() - xi).max() < 1e-9
    a = rng.normal(0, 1, (200, 6))
    assert np.abs(X.adjT(torch.from_numpy(a)).numpy() - ose3.se3_adjT(Xo, a)).max() < 1e-12
    assert np.abs(X.adj(torch.from_numpy(a)).numpy() - ose3.se3_adj(Xo, a)).max() < 1e-12
    assert np.abs((X * X.inv()).data.numpy() - ose3.se3_identity(200, np.float64)).max() < 1e-12
    q = SO3.exp(torch.from_numpy(xi[:, 3:]))
    assert np.abs(q.data.numpy() - ose3.so3_exp(xi[:, 3:])).max() < 1e-12
    p = rng.normal(0, 1, (200, 3))
    assert np.abs(q.act(torch.from_numpy(p)).numpy() - ose3.so3_act(ose3.so3_exp(xi[:, 3:]), p)).max() < 1e-12


def test_out_of_scope_submodules_exist_and_raise():
    import pytest
    from vipe_amd import ext
    for name in ["droid_net_ext", "slam_ext", "lietorch_ext", "scatter_ext", "corr_ext", "utils_ext", "grounding_dino_ext"]:
        assert hasattr(ext, name)
    with pytest.raises(NotImplementedError):
        ext.grounding_dino_ext.ms_deform_attn_forward()
    with pytest.raises(RuntimeError):  # implemented (HIP kernel), device tensors only - like the reference's CHECK_CUDA
        ext.utils_ext.nearest_neighbours(torch.zeros(4, 2), torch.zeros(4, 2), 1)


def test_scatter_host_path_and_autograd():
    """The Python-level scatter API on CPU tensors runs the library's host loop (`vipe_scatter_host`) under the same
    autograd Function as the device kernel: values and adjoints (scatter.cpp:38-201) vs torch.scatter_reduce."""
    from vipe_amd.ext import scatter
    torch.manual_seed(0)
    src = torch.randn(3, 11, 4, dtype=torch.float64, requires_grad=True)
    idx = torch.randint(0, 5, (11,))
    full = idx.view(1, -1, 1).expand(3, 11, 4)
    used = torch.zeros(6, dtype=torch.bool)
    used[idx] = True
    for red, tr, init in (("sum", "sum", 0.0), ("mean", "mean", 0.0), ("mul", "prod", 1.0), ("min", "amin", float("inf")),
                          ("max", "amax", float("-inf"))):
        out = scatter.scatter(src, idx, dim=1, dim_size=6, reduce=red)
        ref = torch.full((3, 6, 4), init, dtype=torch.float64).scatter_reduce(1, full, src, tr, include_self=tr in ("sum", "prod"))
        assert torch.allclose(out[:, used], ref[:, used]), red
        go = torch.randn_like(out)
        (g1,) = torch.autograd.grad(out, src, go, retain_graph=True)
        (g2,) = torch.autograd.grad(ref, src, go)
        assert torch.allclose(g1, g2), red
    assert scatter.scatter_add is scatter.scatter_sum
    assert scatter.scatter_sum(src, idx, 1).shape == (3, int(idx.max()) + 1, 4)  # dim_size from the index
    assert torch.autograd.gradcheck(lambda s: scatter.scatter_mean(s, idx, 1, None, 6), (src,))
    import pytest
    with pytest.raises(_lib.VipeError):
        scatter.scatter_sum(src.detach(), idx + 100, 1, None, 6)  # out-of-range index is an error, not a stray write
    # an index with fewer dims than `dim` (scatter.cpp:19-26 appends trailing dims and broadcasts): index [1,1], dim 2
    s3 = torch.arange(5.0, dtype=torch.float64).view(1, 1, 5)
    assert torch.equal(scatter.scatter_sum(s3, torch.zeros(1, 1, dtype=torch.long), 2), torch.full((1, 1, 1), 10.0, dtype=torch.float64))


def test_corr_ext_host_path_against_torch_autograd():
    """corr_ext on CPU tensors (the reference has a CPU implementation, correlation_sampler.cpp:44-58): forward vs an
    explicit torch formulation of the sampler, backward vs autograd of that formulation."""
    import torch.nn.functional as F
    from vipe_amd.ext import corr_ext
    torch.manual_seed(0)
    B, C, H, W = 2, 5, 9, 11
    a = torch.randn(B, C, H, W, requires_grad=True)
    b = torch.randn(B, C, H, W, requires_grad=True)
    args = dict(kH=1, kW=1, patchH=5, patchW=5, padH=0, padW=0, dilH=1, dilW=1, dil_patchH=2, dil_patchW=1, dH=1, dW=1)
    out = corr_ext.forward(a.detach(), b.detach(), *args.values())
    rH, rW = args["dil_patchH"] * (args["patchH"] - 1) // 2, args["dil_patchW"] * (args["patchW"] - 1) // 2
    bp = F.pad(b, (rW, rW, rH, rH))
    ref = torch.stack([torch.stack([(a * bp[:, :, ph * args["dil_patchH"]:ph * args["dil_patchH"] + H,
                                              pw * args["dil_patchW"]:pw * args["dil_patchW"] + W]).sum(1)
                                    for pw in range(args["patchW"])], 1) for ph in range(args["patchH"])], 1)
    assert out.shape == ref.shape == (B, 5, 5, H, W) and torch.allclose(out, ref, atol=1e-5)
    go = torch.randn_like(ref)
    g1, g2 = corr_ext.backward(a.detach(), b.detach(), go, *args.values())
    r1, r2 = torch.autograd.grad(ref, (a, b), go)
    assert torch.allclose(g1, r1, atol=1e-5) and torch.allclose(g2, r2, atol=1e-5)


def _corr_sampler_cases():
    import os
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py")).read()
    ns = {}
    exec(src[src.index("CORR_SAMPLER_CASES = {"):src.index("def gen_corr_sampler")], ns)
    return ns["CORR_SAMPLER_CASES"]


def test_corr_ext_host_path_matches_the_reference_cpu_implementation(golden_dir):
    """`corr_ext` on CPU tensors vs outputs of the reference's own CPU sampler (csrc/corr_ext/correlation.cpp, compiled
    in the build container by oracle/build_ref.py, frozen by make_golden.gen_corr_sampler): forward and backward, four
    geometries.  Where oracle/_ref/ref_corr.so is present (build container, and it travels to the GPU box) the compiled
    reference is also called directly on fresh inputs."""
    import os
    from vipe_amd.ext import corr_ext
    G = np.load(os.path.join(golden_dir, "corr_sampler_reference.npz"))
    for name, (shape, geom) in _corr_sampler_cases().items():
        a, b, go = (torch.from_numpy(G[f"{name}/{k}"]) for k in ("a", "b", "grad_out"))
        out = corr_ext.forward(a, b, *geom)
        assert out.shape == G[f"{name}/out"].shape and np.allclose(out.numpy(), G[f"{name}/out"], atol=1e-5), name
        g1, g2 = corr_ext.backward(a, b, go, *geom)
        assert np.allclose(g1.numpy(), G[f"{name}/grad1"], atol=1e-5) and np.allclose(g2.numpy(), G[f"{name}/grad2"], atol=1e-5), name
    from oracle.build_ref import load_ref_corr
    ref = load_ref_corr()
    if ref is not None:
        gen = torch.Generator().manual_seed(3)
        a, b = torch.randn(2, 7, 12, 10, generator=gen), torch.randn(2, 7, 12, 10, generator=gen)
        geom = (3, 1, 5, 3, 1, 0, 1, 1, 2, 2, 1, 1)
        want = ref.forward(a, b, *geom)
        assert torch.allclose(corr_ext.forward(a, b, *geom), want, atol=1e-5)
        go = torch.randn(want.shape, generator=gen)
        for x, y in zip(corr_ext.backward(a, b, go, *geom), ref.backward(a, b, go, *geom)):
            assert torch.allclose(x, y, atol=1e-5)


def test_gate_state_piece_validates_its_job_without_a_gpu():
    """`vipe_update_gate_state_piece` (the vipe_overlap_fn the BA calls): argument checks happen before any launch."""
    import ctypes
    from vipe_amd import _lib
    L = _lib.lib()
    Job = _lib.parse_struct("vipe_gate_state_job")
    Buf = _lib.parse_struct("vipe_update_buffers")
    Wt = _lib.parse_struct("vipe_update_weights")
    assert L.vipe_update_gate_state_piece(None, 0, 1, None) == -1
    job, buf, wt = Job(), Buf(), Wt()
    buf.E, buf.H, buf.W = 6, 8, 64
    buf.pgate = buf.pzr = 0x1000
    job.weights, job.buffers, job.net = ctypes.addressof(wt), ctypes.addressof(buf), 0x1000
    assert L.vipe_update_gate_state_piece(ctypes.addressof(job), 3, 3, None) == -1   # piece out of range
    bounds = (ctypes.c_int * 3)(0, 7, 6)                                              # not ascending / beyond E
    job.bounds, job.n_bounds = ctypes.addressof(bounds), 3
    assert L.vipe_update_gate_state_piece(ctypes.addressof(job), 0, 2, None) == -1
    bounds2 = (ctypes.c_int * 3)(0, 6, 6)                                             # empty second piece: nothing to do
    job.bounds = ctypes.addressof(bounds2)
    assert L.vipe_update_gate_state_piece(ctypes.addressof(job), 1, 2, None) == 0


def _sampler_reference(a, b, patch, dil_patch):
    import torch.nn.functional as F
    H, W = a.shape[-2:]
    rH, rW = dil_patch[0] * (patch[0] - 1) // 2, dil_patch[1] * (patch[1] - 1) // 2
    bp = F.pad(b, (rW, rW, rH, rH))
    return torch.stack([torch.stack([(a * bp[:, :, ph * dil_patch[0]:ph * dil_patch[0] + H,
                                             pw * dil_patch[1]:pw * dil_patch[1] + W]).sum(1)
                                     for pw in range(patch[1])], 1) for ph in range(patch[0])], 1)


def test_spatial_correlation_sampler_module_and_autograd_on_cpu():
    """`vipe.ext.corr.SpatialCorrelationSampler` / `spatial_correlation_sample` (spatial_correlation_sampler.py:13-126):
    values and gradients through autograd vs an explicit torch formulation; ints and pairs as size arguments."""
    from vipe_amd.ext.corr import SpatialCorrelationSampler, spatial_correlation_sample
    torch.manual_seed(3)
    a = torch.randn(2, 4, 8, 10, requires_grad=True)
    b = torch.randn(2, 4, 8, 10, requires_grad=True)
    out = SpatialCorrelationSampler(kernel_size=1, patch_size=(3, 5), stride=1, padding=0, dilation=1, dilation_patch=(2, 1))(a, b)
    ref = _sampler_reference(a, b, (3, 5), (2, 1))
    assert out.shape == (2, 3, 5, 8, 10) and torch.allclose(out, ref, atol=1e-5)
    go = torch.randn_like(ref)
    g = torch.autograd.grad(out, (a, b), go)
    r = torch.autograd.grad(ref, (a, b), go)
    assert torch.allclose(g[0], r[0], atol=1e-5) and torch.allclose(g[1], r[1], atol=1e-5)
    out2 = spatial_correlation_sample(a.detach(), b.detach(), patch_size=3)
    assert torch.allclose(out2, _sampler_reference(a.detach(), b.detach(), (3, 3), (1, 1)), atol=1e-5)


def test_motion_filter_sparse_track_score():
    """motion_filter.py:112-135 on a scripted tracker: mean keypoint displacement per view summed, + 100 when more than
    20 % of the previous frame's tracks are gone, nan (never above a threshold) without common keypoints."""
    from vipe_amd.slam.motion_filter import MotionFilter

    class Tracker:
        enabled = True

        def __init__(self):
            self.obs = {}  # frame -> {kp: uv}

        def get_correspondences(self, view_idx, a, b):
            return torch.tensor(sorted(set(self.obs[a]) & set(self.obs[b])), dtype=torch.long)

        def get_observations(self, view_idx, frame, kp):
            return torch.tensor([self.obs[frame][int(k)] for k in kp], dtype=torch.float32).reshape(-1, 2)

    tr = Tracker()
    tr.obs[0] = {k: (10.0 * k, 5.0) for k in range(10)}
    tr.obs[1] = {k: (10.0 * k + 3.0, 9.0) for k in range(10)}       # all ten tracks moved by (3, 4): 5 px
    tr.obs[2] = {k: (10.0 * k + 6.0, 13.0) for k in range(7)}       # three of ten lost: 30 % > 20 %
    tr.obs[3] = {k + 100: (1.0, 1.0) for k in range(4)}             # nothing in common with the keyframe
    mf = MotionFilter(None, sparse_tracks=tr, thresh=2.4, device=torch.device("cpu"))
    mf.last_kf_frame_idx, mf.last_n_sparse_tracks = 0, 0
    mf.current_frame_idx = 1
    assert abs(mf._sparse_motion_score(1) - 5.0) < 1e-6 and mf.last_n_sparse_tracks == 10
    mf.current_frame_idx = 2
    assert abs(mf._sparse_motion_score(1) - 110.0) < 1e-5 and mf.last_n_sparse_tracks == 7
    mf.current_frame_idx = 3
    s3 = mf._sparse_motion_score(1)  # 7 -> 0 tracks: + 100 on top of a nan mean
    assert s3 != s3 and not (s3 > 4.8) and mf.last_n_sparse_tracks == 0
    tr.enabled = False
    assert mf._sparse_motion_score(1) == 0.0


def test_vipe_ext_module_binds_like_the_reference_loader():
    """vipe/ext/__init__.py:24-46 of the reference: `import vipe_ext as _C`, then seven attribute reads.  With this
    repository on the path the unmodified loader finds the package; the calls below go through the C ABI (host loop)."""
    import vipe_ext as _C
    droid_net_ext = _C.droid_net_ext
    grounding_dino_ext = _C.grounding_dino_ext
    utils_ext = _C.utils_ext
    slam_ext = _C.slam_ext
    scatter_ext = _C.scatter_ext
    lietorch_ext = _C.lietorch_ext
    corr_ext = _C.corr_ext
    for mod, names in ((droid_net_ext, ("corr_index_forward", "corr_index_backward", "altcorr_forward", "altcorr_backward")),
                       (slam_ext, ("ba", "frame_distance", "projmap", "depth_filter", "iproj")),
                       (lietorch_ext, ("expm", "logm", "inv", "mul", "adj", "adjT", "act", "act4", "as_matrix", "projector",
                                       "Jinv", "expm_backward", "mul_backward", "act4_backward")),
                       (scatter_ext, ("scatter_sum", "scatter_mean", "scatter_mul", "scatter_min", "scatter_max")),
                       (corr_ext, ("forward", "backward")), (utils_ext, ("nearest_neighbours",)),
                       (grounding_dino_ext, ("ms_deform_attn_forward", "ms_deform_attn_backward"))):
        for n in names:
            assert callable(getattr(mod, n)), (mod, n)
    xi = torch.tensor([[0.1, -0.2, 0.3, 0.02, 0.01, -0.03]], dtype=torch.float64)
    X = _C.lietorch_ext.expm(3, xi)  # SE3 group id 3, as the reference's group classes call it (group_ops.py)
    assert np.abs(X.numpy() - ose3.se3_exp(xi.numpy())).max() < 1e-12
    assert np.abs(_C.lietorch_ext.logm(3, X).numpy() - xi.numpy()).max() < 1e-12
    import pytest
    with pytest.raises(NotImplementedError):
        _C.grounding_dino_ext.ms_deform_attn_forward()
