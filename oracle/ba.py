"""Dense bundle adjustment on SE3 (+) inverse depth: block Gauss-Newton with Schur.

ORACLE (test infrastructure). numpy, dtype-generic (float64 = "exact" checker,
float32 = reference arithmetic). Restates the LIVE Python BA of the reference:
  vipe/slam/components/buffer.py:373-525  (problem set-up, fixed sets, damping)
  vipe/slam/ba/terms.py:94-303            (DenseDepthFlowTerm, DispSensRegularizationTerm)
  vipe/slam/ba/solver.py:117-197          (normal equations, Schur, back-substitution)
  vipe/slam/maths/matrix.py:179-192,346-358 (damping), retractor.py:22-62
The reduced system is assembled densely (<= 6N + V(1+D) + 6V unknowns); the
reference solves it with fp32 SuperLU (solver.py:33-44) - here numpy.linalg.solve
in the working dtype (or scipy spsolve in fp32 when solver="spsolve").
Per-pixel disparity damping is applied by frame id (the reference's searchsorted
mapping, matrix.py:353-358, is only sound when no disparity frame is fixed).
"""

import numpy as np

from . import geom, se3


def _merge(blocks):
    """Sum Jacobians that hit the same variable block (e.g. qi == qj for intrinsics)."""
    out = {}
    for key, J in blocks:
        out[key] = out[key] + J if key in out else J
    return out


def bundle_adjustment(
    poses, disps, disps_sens, intrinsics, rig, target, weight, disp_damping, ii, jj, t0, t1, n_iters,
    pose_damping, pose_ep, motion_only=False, limited_disp=False, optimize_intrinsics=False,
    optimize_rig_rotation=False, model="pinhole", alpha=0.001, dtype=np.float64, solver="dense",
    cross_view_idx=None, return_debug=False, weight_scale=0.001, intrinsics_factor=8.0,
):
    """buffer.py:373-525.

    poses [Nbuf,7]; disps, disps_sens, disp_damping [Nbuf,V,ht,wd]; intrinsics [V,4+D] full-res;
    rig [V,7]; target, weight [E*V,P,2]; ii, jj [E].  Returns updated copies
    (poses, disps, intrinsics, rig) and optionally a list of per-iteration debug dicts.
    """
    dt = np.dtype(dtype)
    poses = np.array(poses, dtype=dt)
    disps = np.array(disps, dtype=dt)
    intrinsics = np.array(intrinsics, dtype=dt)
    rig = np.array(rig, dtype=dt)
    Nbuf, V, ht, wd = disps.shape
    P = ht * wd
    dflat = disps.reshape(Nbuf * V, P)
    sens = np.asarray(disps_sens, dtype=dt).reshape(Nbuf * V, P)
    eta = np.asarray(disp_damping, dtype=dt).reshape(Nbuf * V, P)
    ii = np.asarray(ii, dtype=np.int64)
    jj = np.asarray(jj, dtype=np.int64)
    assert t0 <= t1
    pi, qi, di, pj, qj, dj = geom.expand_edge_multiview(ii, jj, V, cross_view_idx)
    M = len(pi)
    target = np.asarray(target, dtype=dt).reshape(M, P, 2)
    wgt = (np.asarray(weight, dtype=dt) * dt.type(weight_scale)).reshape(M, P, 2)  # buffer.py:413
    D = intrinsics.shape[1] - 4

    # ---- fixed sets (buffer.py:462-465, 490-493; terms.py:69-72)
    pi_unique = np.unique(ii)
    if t0 < t1:
        fixed_pose = set(pi_unique[(pi_unique < t0) | (pi_unique >= t1)].tolist())
        all_pose_fixed = False
    else:
        fixed_pose, all_pose_fixed = set(), True
    di_unique = np.unique(di)
    if motion_only:
        fixed_disp = set(di_unique.tolist())
    elif limited_disp:
        fixed_disp = set(di[(pi < t0) | (pi >= t1)].tolist())
    else:
        fixed_disp = set()
    free_disp = [int(k) for k in di_unique if int(k) not in fixed_disp]
    # sensor prior frames (buffer.py:470-479): float32 sum > 0
    sens_frames = [k for k in free_disp if float(np.asarray(disps_sens, dtype=np.float32).reshape(Nbuf * V, P)[k].sum()) > 0.0]

    # ---- regular variable blocks: poses, intrinsics (per view), rig (per view, view 0 fixed)
    pose_ids = [] if all_pose_fixed else sorted(set(pi.tolist() + pj.tolist()) - fixed_pose)
    blocks = [("pose", p, 6) for p in pose_ids]
    if optimize_intrinsics:
        blocks += [("intr", q, 1 + D) for q in range(V)]
    if optimize_rig_rotation:
        blocks += [("rig", q, 6) for q in range(1, V)]  # buffer.py:506
    off, n = {}, 0
    for kind, idx, dim in blocks:
        off[(kind, idx)] = (n, dim)
        n += dim

    debug = []
    for _ in range(n_iters):
        intr8 = geom.scaled_intrinsics(intrinsics, 1.0 / intrinsics_factor, model)  # terms.py:182
        g = geom.reproject(poses, dflat.reshape(Nbuf * V, ht, wd), intr8, rig, pi, pj, qi, qj, di, model,
                           jacobian=True, jacobian_f=optimize_intrinsics)
        r = (g["coords"].reshape(M, P, 2) - target)  # terms.py:242
        w = g["valid"].reshape(M, P, 1) * wgt  # terms.py:195
        Ji = g["Ji"].reshape(M, P, 2, 6)
        Jj = g["Jj"].reshape(M, P, 2, 6)
        Jz = g["Jz"].reshape(M, P, 2)

        H = np.zeros((n, n), dtype=dt)
        v = np.zeros(n, dtype=dt)
        Eblk = {}  # (block key, frame k) -> [dim, P]
        C = {k: np.zeros(P, dtype=dt) for k in free_disp}
        wv = {k: np.zeros(P, dtype=dt) for k in free_disp}
        for e in range(M):
            loc = []
            if ("pose", int(pi[e])) in off:
                loc.append((("pose", int(pi[e])), Ji[e]))
            if ("pose", int(pj[e])) in off:
                loc.append((("pose", int(pj[e])), Jj[e]))
            if optimize_intrinsics:
                sc = dt.type(1.0 / intrinsics_factor)  # terms.py:224-227 (J_scale)
                loc.append((("intr", int(qi[e])), g["Jfi"][e].reshape(P, 2, 1 + D) * sc))
                loc.append((("intr", int(qj[e])), g["Jfj"][e].reshape(P, 2, 1 + D) * sc))
            if optimize_rig_rotation:
                if ("rig", int(qi[e])) in off:
                    loc.append((("rig", int(qi[e])), -Ji[e]))  # geom.py:292-294
                if ("rig", int(qj[e])) in off:
                    loc.append((("rig", int(qj[e])), -Jj[e]))
            loc = _merge(loc)
            keys = list(loc.keys())
            for a in keys:
                oa, da = off[a]
                Ja_w = loc[a] * w[e][:, :, None]
                v[oa:oa + da] += -np.einsum("pcd,pc->d", Ja_w, r[e])  # nwjtr, terms.py:66-67
                for b in keys:
                    ob, db = off[b]
                    H[oa:oa + da, ob:ob + db] += np.einsum("pcd,pcf->df", Ja_w, loc[b])
            k = int(di[e])
            if k in C:
                wz = w[e] * Jz[e]
                C[k] += np.sum(wz * Jz[e], axis=1)
                wv[k] += -np.sum(wz * r[e], axis=1)
                for a in keys:
                    Eak = np.einsum("pcd,pc->dp", loc[a], wz)
                    Eblk[(a, k)] = Eblk[(a, k)] + Eak if (a, k) in Eblk else Eak

        # sensor-depth prior (terms.py:246-303)
        for k in sens_frames:
            C[k] += dt.type(alpha)
            wv[k] += -dt.type(alpha) * (dflat[k] - sens[k])

        # damping (solver.py:161-164; matrix.py:179-192, 346-358)
        for kind, idx, dim in blocks:
            o, _d = off[(kind, idx)]
            lam, ep = {"pose": (pose_damping, pose_ep), "intr": (1e-6, 1e-6), "rig": (1e-4, 1e-4)}[kind]
            for a in range(o, o + dim):
                H[a, a] += dt.type(ep) + dt.type(lam) * H[a, a]
        for k in free_disp:
            C[k] += dt.type(1e-7) + (dt.type(0.2) * eta[k] + dt.type(1e-7))  # buffer.py:482-489

        # Schur (solver.py:170-178)
        S = H.copy()
        gv = v.copy()
        by_frame = {}
        for (a, k) in Eblk:
            by_frame.setdefault(k, []).append(a)
        for k, members in by_frame.items():
            Q = 1.0 / C[k]
            for a in members:
                oa, da = off[a]
                EQ = Eblk[(a, k)] * Q[None]
                gv[oa:oa + da] -= EQ @ wv[k]
                for b in members:
                    ob, db = off[b]
                    S[oa:oa + da, ob:ob + db] -= EQ @ Eblk[(b, k)].T

        if n > 0:
            if solver == "spsolve":  # solver.py:33-44: fp32 COO -> CSR -> SuperLU
                from scipy.sparse import coo_matrix
                from scipy.sparse.linalg import spsolve
                nz = np.nonzero(S)
                dx = spsolve(coo_matrix((S[nz].astype(np.float32), nz), shape=S.shape).tocsr(),
                             gv.astype(np.float32)).astype(dt)
            else:
                dx = np.linalg.solve(S, gv)
        else:
            dx = np.zeros(0, dtype=dt)

        # back-substitution (solver.py:182-183)
        dz = {}
        for k in free_disp:
            rhs = wv[k].copy()
            for a in by_frame.get(k, []):
                oa, da = off[a]
                rhs -= Eblk[(a, k)].T @ dx[oa:oa + da]
            dz[k] = rhs / C[k]

        # retraction (retractor.py:27-62)
        for kind, idx, dim in blocks:
            o, _d = off[(kind, idx)]
            step = dx[o:o + dim]
            if kind == "pose":
                poses[idx] = se3.se3_retr(poses[idx], step)
            elif kind == "rig":
                s2 = step.copy()
                s2[:3] = 0
                rig[idx] = se3.se3_retr(rig[idx], s2)
            elif kind == "intr":
                # one block per view here (the reference broadcasts only when a single block exists)
                intrinsics[idx, :2] += step[0]
                if D > 0:
                    intrinsics[idx, 4:] += step[1:] * dt.type(0.01)
        for k in free_disp:
            step = np.where(dz[k] > 10, np.zeros_like(dz[k]), dz[k])  # retractor.py:41
            dflat[k] += step
        if return_debug:
            debug.append({"H": H, "v": v, "S": S, "g": gv, "dx": dx, "dz": dz, "C": C, "w": wv,
                          "blocks": blocks, "off": off, "r": r, "wgt": w})

    np.maximum(dflat, dt.type(1e-3), out=dflat)  # buffer.py:525 (whole buffer)
    out = (poses, dflat.reshape(Nbuf, V, ht, wd), intrinsics, rig)
    return out + (debug,) if return_debug else out


def energy(poses, disps, intrinsics, rig, target, weight, ii, jj, model="pinhole", dtype=np.float64,
           weight_scale=0.001):
    """terms.py:78-79: sum w r^2 of the dense flow term."""
    dt = np.dtype(dtype)
    disps = np.asarray(disps, dtype=dt)
    Nbuf, V, ht, wd = disps.shape
    P = ht * wd
    pi, qi, di, pj, qj, _ = geom.expand_edge_multiview(ii, jj, V)
    intr8 = geom.scaled_intrinsics(np.asarray(intrinsics, dtype=dt), 1.0 / 8.0, model)
    g = geom.reproject(np.asarray(poses, dtype=dt), disps.reshape(Nbuf * V, ht, wd), intr8,
                       np.asarray(rig, dtype=dt), pi, pj, qi, qj, di, model)
    r = g["coords"].reshape(len(pi), P, 2) - np.asarray(target, dtype=dt).reshape(len(pi), P, 2)
    w = g["valid"].reshape(len(pi), P, 1) * np.asarray(weight, dtype=dt).reshape(len(pi), P, 2) * weight_scale
    return float(np.sum(w * r * r))
