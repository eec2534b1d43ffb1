"""slam_ext: projective geometry + dense BA (reference: csrc/slam_ext/slam.cpp:31-37)."""

import ctypes

import torch

from .._lib import CAMERA_CODE, BAParams, check, check_gpu_contig, lib, ptr, require, stream_ptr

_WS = {}


def _workspace(device, nbytes):
    """Grow-only per-device workspace (allocated outside the timed / captured region after the first call)."""
    key = (device.type, device.index)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def _i64(t):
    require(t.dtype == torch.int64, "index tensors must be int64")
    return t


def reproject(poses, disps, intrinsics, rig, pi, qi, pj, qj, di, camera="pinhole", intr_factor=8.0, want_valid=True):
    """GraphBuffer.reproject_dense_disp core (buffer.py:527-548). disps [NV,ht,wd]; -> coords [M,ht,wd,2], valid [M,ht,wd,1]."""
    check_gpu_contig(poses, disps, intrinsics, rig, pi, qi, pj, qj, di)
    M = pi.shape[0]
    _, ht, wd = disps.shape
    coords = torch.empty((M, ht, wd, 2), dtype=torch.float32, device=poses.device)
    valid = torch.empty((M, ht, wd, 1), dtype=torch.float32, device=poses.device) if want_valid else None
    check(lib().vipe_reproject(ptr(poses), ptr(disps), ptr(intrinsics), ptr(rig), ptr(_i64(pi)), ptr(_i64(qi)),
                               ptr(_i64(pj)), ptr(_i64(qj)), ptr(_i64(di)), ptr(coords), ptr(valid), M, ht, wd,
                               rig.shape[0], CAMERA_CODE[camera], float(intr_factor), stream_ptr(poses)), "reproject")
    return coords, valid


def reproject_motion(poses, disps, intrinsics, rig, pi, qi, pj, qj, di, target, camera="pinhole", intr_factor=8.0,
                     motn_dtype=torch.float16):
    """coords1 + clamped motion features [M,4,ht,wd] in one launch (factor_graph.py:253-261)."""
    check_gpu_contig(poses, disps, intrinsics, rig, pi, qi, pj, qj, di, target)
    M = pi.shape[0]
    _, ht, wd = disps.shape
    require(target.numel() == M * ht * wd * 2 and target.dtype == torch.float32, "target must be [M,ht,wd,2] float32")
    coords = torch.empty((M, ht, wd, 2), dtype=torch.float32, device=poses.device)
    motn = torch.empty((M, 4, ht, wd), dtype=motn_dtype, device=poses.device)
    code = {torch.float16: 0, torch.float32: 1}[motn_dtype]
    check(lib().vipe_reproject_motion(ptr(poses), ptr(disps), ptr(intrinsics), ptr(rig), ptr(_i64(pi)), ptr(_i64(qi)),
                                      ptr(_i64(pj)), ptr(_i64(qj)), ptr(_i64(di)), ptr(target), ptr(coords), ptr(motn),
                                      M, ht, wd, rig.shape[0], CAMERA_CODE[camera], float(intr_factor), code,
                                      stream_ptr(poses)), "reproject_motion")
    return coords, motn


def reproject_motion_nhwc(poses, disps, intrinsics, rig, pi, qi, pj, qj, di, target, camera="pinhole", intr_factor=8.0,
                          out=None):
    """coords1 + clamped motion features channels-last [M,ht,wd,4] fp16 in one launch.  `out` = (coords, motn) buffers
    to write into (a caller iterating over an unchanged edge set keeps them: stable addresses, no allocation)."""
    check_gpu_contig(poses, disps, intrinsics, rig, pi, qi, pj, qj, di, target)
    M = pi.shape[0]
    _, ht, wd = disps.shape
    require(target.numel() == M * ht * wd * 2 and target.dtype == torch.float32, "target must be [M,ht,wd,2] float32")
    if out is not None:
        coords, motn = out
        require(tuple(coords.shape) == (M, ht, wd, 2) and coords.dtype == torch.float32 and coords.is_contiguous()
                and tuple(motn.shape) == (M, ht, wd, 4) and motn.dtype == torch.float16 and motn.is_contiguous(), "bad out buffers")
    else:
        coords = torch.empty((M, ht, wd, 2), dtype=torch.float32, device=poses.device)
        motn = torch.empty((M, ht, wd, 4), dtype=torch.float16, device=poses.device)
    check(lib().vipe_reproject_motion_nhwc(ptr(poses), ptr(disps), ptr(intrinsics), ptr(rig), ptr(_i64(pi)),
                                           ptr(_i64(qi)), ptr(_i64(pj)), ptr(_i64(qj)), ptr(_i64(di)), ptr(target),
                                           ptr(coords), ptr(motn), M, ht, wd, rig.shape[0], CAMERA_CODE[camera],
                                           float(intr_factor), stream_ptr(poses)), "reproject_motion_nhwc")
    return coords, motn


def update_finish(coords1, dw, mask, target, weight, eta, du, damping):
    """Tail of FactorGraph.update (factor_graph.py:270-276) in one launch: target = coords1 + delta, weight = w with the
    masked source frames zeroed, damping[du] = eta.  coords1 [E,h,w,2], dw [E,h,w,4] f32 (delta | weight), mask [E,h,w]
    bool or None, target / weight [1,E,h,w,2] (written in place), eta [n_src,h,w], du [n_src] int64, damping [*,h,w]."""
    E, ht, wd, _ = coords1.shape
    n_src = 0 if eta is None else int(eta.shape[0])
    require(target.is_contiguous() and weight.is_contiguous() and target.numel() == E * ht * wd * 2, "target / weight: contiguous [1,E,h,w,2]")
    check(lib().vipe_update_finish(ptr(coords1), ptr(dw), ptr(mask), ptr(target), ptr(weight), ptr(eta), ptr(du), ptr(damping),
                                   E, n_src, ht, wd, stream_ptr(coords1)), "update_finish")


BA_OPT_ONE_CHAIN, BA_OPT_GENERAL_ACCUMULATE = 1, 2  # VIPE_BA_OPT_* (include/vipe_amd.h)
PROFILE_EVENTS = None  # measurement aid (bench.py): (event, event[, iteration]) every dense_ba call records around that iteration's accumulate launch
PLAN_REUSE = True  # False: rebuild the edge plan on every call and launch every kernel of both paths (validation aid)


def dense_ba(poses, disps, disps_sens, intrinsics, rig, target, weight, disp_damping, pi, qi, pj, qj, di, t0, t1,
             n_iters, pose_damping, pose_ep, motion_only=False, limited_disp=False, optimize_intrinsics=False,
             optimize_rig_rotation=False, camera="pinhole", alpha=0.001, n_poses=None, want_info=False, state=None,
             plan_key=None, overlap=None, solver_options=0, profile_events=None):
    """Live dense BA (GraphBuffer.bundle_adjustment, buffer.py:373-525), IN PLACE on poses / disps / intrinsics.

    `overlap` = (stream address, vipe_overlap_fn address, user address) or None: independent work of the caller that the
    library enqueues on that stream in max(n_iters, 1) pieces, each behind the start of a Gauss-Newton iteration's
    single-workgroup solve (include/vipe_amd.h, vipe_overlap_fn).  `solver_options`: VIPE_BA_OPT_* bits (BA_OPT_ONE_CHAIN,
    BA_OPT_GENERAL_ACCUMULATE) - the general forms of the specialised kernels, for validating one against the other.
    `profile_events`: see vipe_ba_params.profile_ev0 (the last iteration's accumulate launch between two events).

    poses [>=n_poses,7]; disps, disps_sens, disp_damping [>=n_poses*V,ht,wd] (flattened views);
    target, weight [M,ht*wd,2]; pi..di [M] int64.  `n_poses` bounds the pose/frame indices that occur
    (default: all rows of `poses`); a tight bound keeps the reduced system small."""
    check_gpu_contig(poses, disps, disps_sens, intrinsics, rig, target, weight, disp_damping, pi, qi, pj, qj, di)
    for t in (poses, disps, disps_sens, intrinsics, rig, target, weight, disp_damping):
        require(t.dtype == torch.float32, "dense_ba works in float32 (the reference's BA dtype)")
    V = rig.shape[0]
    n_poses = poses.shape[0] if n_poses is None else int(n_poses)
    require(disps.dim() == 3 and disps.shape[0] >= n_poses * V, "disps must be [n_poses*V,ht,wd]")
    _, ht, wd = disps.shape
    M = pi.shape[0]
    require(target.numel() == M * ht * wd * 2 and weight.numel() == M * ht * wd * 2, "target/weight must be [M,P,2]")
    p = BAParams(n_poses=n_poses, n_views=V, ht=ht, wd=wd, M=M, t0=int(t0), t1=int(t1), n_iters=int(n_iters),
                 pose_damping=float(pose_damping), pose_ep=float(pose_ep), motion_only=int(motion_only),
                 limited_disp=int(limited_disp), optimize_intrinsics=int(optimize_intrinsics),
                 optimize_rig_rotation=int(optimize_rig_rotation), camera=CAMERA_CODE[camera], alpha=float(alpha),
                 weight_scale=0.001, intr_factor=8.0, reuse_plan=0, path_hint=0, solver_options=int(solver_options))
    if overlap is not None:
        p.overlap_stream, p.overlap_fn, p.overlap_user = overlap
    if profile_events is None:
        profile_events = PROFILE_EVENTS
    if profile_events is not None:  # two torch.cuda.Event(enable_timing=True), each recorded once before (so that they exist)
        p.profile_ev0, p.profile_ev1 = (int(e.cuda_event) for e in profile_events[:2])
        p.profile_iter = int(profile_events[2]) if len(profile_events) > 2 else -1
    L = lib()
    nbytes = L.vipe_dense_ba_workspace_bytes(ctypes.byref(p))
    require(nbytes > 0, "bad BA parameters")
    key = None
    if state is None:
        ws = _workspace(poses.device, nbytes)
    else:
        # `state`: a dict owned by ONE caller (a FactorGraph) holding a private workspace.  When that caller vouches, through
        # `plan_key`, that the index arrays are the ones of its previous call, the plan left in the workspace is reused
        # (vipe_ba_params.reuse_plan); everything else that shapes the plan is part of the key checked here.
        ws = state.get("ws")
        if ws is None or ws.numel() < nbytes or ws.device != poses.device:
            ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=poses.device)
            state["ws"], state["key"] = ws, None
        key = None if plan_key is None else (plan_key, ws.data_ptr(), n_poses, V, ht, wd, M, int(t0), int(t1), int(motion_only),
                                             int(limited_disp), int(optimize_intrinsics), int(optimize_rig_rotation), camera,
                                             tuple(int(x.data_ptr()) for x in (pi, qi, pj, qj, di)))
        p.reuse_plan = int(key is not None and state.get("key") == key and PLAN_REUSE)
        # a call that launches nothing (no terms / no iterations) leaves the workspace as it found it: it vouches for nothing
        state["key"] = key if (M > 0 and int(n_iters) > 0) else None
        p.path_hint = _path_hint(state, key)
    learn = state is not None and key is not None and p.path_hint == 0 and "pending" not in state and M > 0 and n_iters > 0 \
        and not torch.cuda.is_current_stream_capturing() and PLAN_REUSE
    info = torch.zeros(8, dtype=torch.int32, device=poses.device) if (want_info or learn) else None
    check(L.vipe_dense_ba(ctypes.byref(p), ptr(poses), ptr(disps), ptr(disps_sens), ptr(intrinsics), ptr(rig),
                          ptr(target), ptr(weight), ptr(disp_damping), ptr(_i64(pi)), ptr(_i64(qi)), ptr(_i64(pj)),
                          ptr(_i64(qj)), ptr(_i64(di)), ptr(ws), ws.numel(), ptr(info), stream_ptr(poses)), "dense_ba")
    if learn:  # read the plan's facts back WITHOUT blocking: pinned buffer + event, consulted by a later call
        host = torch.empty(8, dtype=torch.int32, pin_memory=True)
        host.copy_(info, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        state["pending"] = (key, host, ev)
    return info


def _path_hint(state, key):
    """vipe_ba_params.path_hint for the plan `key` of this caller, or 0 while nothing has been learnt about it yet."""
    if key is None or not PLAN_REUSE:
        return 0
    hints = state.setdefault("hints", {})
    pend = state.get("pending")
    # no event query while a stream is being captured (a query of an event recorded outside the capture is legal in
    # relaxed mode only): the hint simply stays unknown for this call
    if pend is not None and not torch.cuda.is_current_stream_capturing() and pend[2].query():
        k, host, _ = pend
        del state["pending"]
        if len(hints) > 64:
            hints.clear()
        degree, solved = int(host[6]), int(host[5])  # solved: 0 global-memory Cholesky, 1 LDS band solver, 2 LDS dense solver
        if degree > 0:  # AM_DMAX = 6 in csrc/ba.hip: the matrix-core accumulate kernel covers source degrees up to 6
            hints[k] = (1 if degree <= 6 else 2) | {0: 8 | 16, 1: 4 | 16, 2: 4 | 8}.get(solved, 0)
    return hints.get(key, 0)


# ------------------------------------------------------------------ reference-named entry points (slam.cpp:31-37)


def frame_distance(poses, disps, intrinsics, pi, pj, qi, qj, di, beta):
    """geom_kernels.cu:1406-1434. poses [NV,7], disps [NV,ht,wd], intrinsics [V,4] -> dist [M]."""
    check_gpu_contig(poses, disps, intrinsics, pi, pj, qi, qj, di)
    M = pi.shape[0]
    _, ht, wd = disps.shape
    dist = torch.empty(M, dtype=torch.float32, device=poses.device)
    check(lib().vipe_frame_distance(ptr(poses), ptr(disps), ptr(intrinsics), ptr(_i64(pi)), ptr(_i64(pj)), ptr(_i64(qi)),
                                    ptr(_i64(qj)), ptr(_i64(di)), ptr(dist), M, ht, wd, float(beta), stream_ptr(poses)),
          "frame_distance")
    return dist


def frame_distance_rig(poses, rig, disps, intrinsics, pi, qi, pj, qj, beta, bidirectional=True, intr_factor=8.0):
    """[fused] buffer.py:550-593 in one launch: poses [N,7], rig [V,7], disps [N*V,ht,wd], intrinsics [V,4|5] at full
    resolution, (pi, qi) -> (pj, qj) [M] int64 -> dist [M] (`vipe_frame_distance_rig`)."""
    check_gpu_contig(poses, rig, disps, intrinsics, pi, qi, pj, qj)
    M = pi.shape[0]
    _, ht, wd = disps.shape
    dist = torch.empty(M, dtype=torch.float32, device=poses.device)
    check(lib().vipe_frame_distance_rig(ptr(poses), ptr(rig), ptr(disps), ptr(intrinsics), int(intrinsics.shape[1]),
                                        float(intr_factor), ptr(_i64(pi)), ptr(_i64(qi)), ptr(_i64(pj)), ptr(_i64(qj)), ptr(dist),
                                        M, int(rig.shape[0]), ht, wd, float(beta), int(bool(bidirectional)), stream_ptr(poses)),
          "frame_distance_rig")
    return dist


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """geom_kernels.cu:1462-1486 -> counter [num,ht,wd] float32."""
    check_gpu_contig(poses, disps, intrinsics, ix, thresh)
    n, ht, wd = disps.shape
    num = ix.shape[0]
    counter = torch.zeros((num, ht, wd), dtype=torch.float32, device=poses.device)
    check(lib().vipe_depth_filter(ptr(poses), ptr(disps), ptr(intrinsics), ptr(_i64(ix)), ptr(thresh), ptr(counter), n,
                                  num, ht, wd, stream_ptr(poses)), "depth_filter")
    return counter


def projmap(poses, disps, intrinsics, ii, jj):
    """geom_kernels.cu:1436-1460 -> [coords [E,ht,wd,3], valid [E,ht,wd,1]]."""
    check_gpu_contig(poses, disps, intrinsics, ii, jj)
    E = ii.shape[0]
    _, ht, wd = disps.shape
    coords = torch.zeros((E, ht, wd, 3), dtype=torch.float32, device=poses.device)
    valid = torch.zeros((E, ht, wd, 1), dtype=torch.float32, device=poses.device)
    check(lib().vipe_projmap(ptr(poses), ptr(disps), ptr(intrinsics), ptr(_i64(ii)), ptr(_i64(jj)), ptr(coords),
                             ptr(valid), E, ht, wd, stream_ptr(poses)), "projmap")
    return [coords, valid]


def iproj(poses, disps, intrinsics):
    """geom_kernels.cu:1488-1507 -> points [n,ht,wd,3]."""
    check_gpu_contig(poses, disps, intrinsics)
    n, ht, wd = disps.shape
    pts = torch.zeros((n, ht, wd, 3), dtype=torch.float32, device=poses.device)
    check(lib().vipe_iproj(ptr(poses), ptr(disps), ptr(intrinsics), ptr(pts), n, ht, wd, stream_ptr(poses)), "iproj")
    return pts


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only):
    """DROID BA signature (slam.cpp:31, geom_kernels.cu:1273-1404) - dormant in the reference."""
    check_gpu_contig(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj)
    n, ht, wd = disps.shape
    E = ii.shape[0]
    L = lib()
    nbytes = L.vipe_ba_workspace_bytes(n, ht, wd, E)
    check(min(nbytes, 0), "ba")
    ws = _workspace(poses.device, nbytes)
    dx = torch.zeros((t1 - t0, 6), dtype=torch.float32, device=poses.device)
    dz = torch.zeros((eta.shape[0], ht * wd), dtype=torch.float32, device=poses.device)
    check(L.vipe_ba(ptr(poses), ptr(disps), ptr(intrinsics), ptr(disps_sens), ptr(targets), ptr(weights), ptr(eta),
                    ptr(_i64(ii)), ptr(_i64(jj)), n, ht, wd, E, eta.shape[0], int(t0), int(t1), int(iterations),
                    float(lm), float(ep), int(motion_only), ptr(dx), ptr(dz), ptr(ws), ws.numel(),
                    stream_ptr(poses)), "ba")
    return [dx, dz]
