"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference's dormant DROID bundle adjustment,
`slam_ext.ba` = `ba_cuda` (csrc/slam_ext/geom_kernels.cu:1273-1404) with its kernels
`projective_transform_kernel` (:178-432), `EEt6x6_kernel` / `Ev6x1_kernel` / `EvT6x1_kernel` (:994-1098),
`accum_cuda` (:961-992), `SparseBlock` (:1100-1198), `schur_block` (:1200-1271), `pose_retr_kernel` (:901-931),
`disp_retr_kernel` (:933-946).

PARITY UNPINNED: the reference implementation is CUDA + Eigen and cannot be built or run in this environment, it is
never called by the reference's Python and the reference holds no test or fixture for it.  This file
follows the source text, including its quirks:
  * MIN_DEPTH 0.25 and target-side-only validity; weights 0.001 * w; residual = target - projection (:301-309);
  * stereo terms (ii == jj): fixed baseline (-0.1, 0, 0), they only contribute to the disparity system (:222-233, 328, 364);
  * pose blocks with an index < t0 are dropped from the system (`update_lhs` / `update_rhs`: i >= 0 && j >= 0);
  * depth prior: per-pixel mask m = disps_sens > 0: C += m * 0.05 + (1 - m) * eta, w -= m * 0.05 * (d - d_sens) (:1359-1369);
  * damping is applied to the diagonal of the REDUCED system, L.diag += ep + lm * L.diag (:1175-1176);
  * `EvT6x1_kernel` skips blocks whose pose index relative to t0 is <= 0 (:1085): pose t0 never enters the disparity
    back-substitution;
  * no dz rejection, no clamp of the disparities.
float64 arithmetic (the reference: fp32 kernels, fp64 sparse Cholesky)."""
import numpy as np

from . import se3

MIN_DEPTH = 0.25


def _adjT(t, R, X):
    """adjSE3 (:95-110): Adj(T)^T X for T = (R, t); X [..., 6]."""
    a, b = X[..., :3], X[..., 3:]
    return np.concatenate([a @ R, (b - np.cross(t, a)) @ R], -1)


def droid_ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep,
             motion_only):
    """poses [N,7], disps / disps_sens [N,ht,wd], intrinsics [4] (at the map's scale), targets / weights [E,2,ht,wd],
    eta [K,ht,wd] (K = number of unique frames in arange(t0,t1) U ii, sorted), ii, jj [E].
    Returns (poses, disps, dx [t1-t0,6], dz [K,ht*wd]) - updated copies and the last iteration's steps."""
    poses = np.array(poses, np.float64)
    disps = np.array(disps, np.float64)
    sens = np.asarray(disps_sens, np.float64)
    fx, fy, cx, cy = [float(x) for x in intrinsics]
    ii, jj = np.asarray(ii, np.int64), np.asarray(jj, np.int64)
    E, (N, ht, wd) = len(ii), disps.shape
    Pn = t1 - t0
    Pp = ht * wd
    ts = np.arange(t0, t1)
    ii_exp, jj_exp = np.concatenate([ts, ii]), np.concatenate([ts, jj])
    kx, kk_exp = np.unique(ii_exp, return_inverse=True)
    K = len(kx)
    u, v = np.meshgrid(np.arange(wd, dtype=np.float64), np.arange(ht, dtype=np.float64))
    u, v = u.reshape(-1), v.reshape(-1)
    eta = np.asarray(eta, np.float64).reshape(K, Pp)
    dx = np.zeros((Pn, 6))
    dz = np.zeros((K, Pp))
    for _ in range(iterations):
        Hs = np.zeros((4, E, 6, 6))
        vs = np.zeros((2, E, 6))
        Eii = np.zeros((E, 6, Pp))
        Eij = np.zeros((E, 6, Pp))
        Cii = np.zeros((E, Pp))
        bz = np.zeros((E, Pp))
        for e in range(E):
            i, j = int(ii[e]), int(jj[e])
            if i == j:
                t, R = np.array([-0.1, 0.0, 0.0]), np.eye(3)
            else:
                T = se3.se3_mul(poses[j][None], se3.se3_inv(poses[i][None]))  # relSE3: G_j G_i^-1
                M = se3.se3_matrix(T)[0]
                R, t = M[:3, :3], M[:3, 3]
            d_i = disps[i].reshape(-1)
            Xi = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
            Xj = Xi @ R.T + d_i[:, None] * t
            x, y, z, h = Xj[:, 0], Xj[:, 1], Xj[:, 2], d_i
            ok = ~(z < MIN_DEPTH)
            d = np.where(ok, 1.0 / np.where(ok, z, 1.0), 0.0)
            d2 = d * d
            w_uv = [np.where(ok, 0.001 * weights[e, c].reshape(-1), 0.0) for c in range(2)]
            r_uv = [targets[e, 0].reshape(-1) - (fx * d * x + cx), targets[e, 1].reshape(-1) - (fy * d * y + cy)]
            zero = np.zeros_like(x)
            Jj_uv = [np.stack([fx * (h * d), zero, fx * (-x * h * d2), fx * (-x * y * d2), fx * (1 + x * x * d2), fx * (-y * d)], -1),
                     np.stack([zero, fy * (h * d), fy * (-y * h * d2), fy * (-1 - y * y * d2), fy * (x * y * d2), fy * (x * d)], -1)]
            Jz_uv = [fx * (t[0] * d - t[2] * (x * d2)), fy * (t[1] * d - t[2] * (y * d2))]
            for c in range(2):
                w, r, Jj, Jz = w_uv[c], r_uv[c], Jj_uv[c], Jz_uv[c]
                Cii[e] += w * Jz * Jz
                bz[e] += w * r * Jz
                if i == j:
                    continue  # stereo: pose terms get zero weight
                Ji = -_adjT(t, R, Jj)
                Jx = np.concatenate([Ji, Jj], -1)  # [P, 12]
                H = np.einsum("p,pn,pm->nm", w, Jx, Jx)
                Hs[0, e] += H[:6, :6]
                Hs[1, e] += H[:6, 6:]
                Hs[2, e] += H[6:, :6]
                Hs[3, e] += H[6:, 6:]
                vs[0, e] += (w * r) @ Ji
                vs[1, e] += (w * r) @ Jj
                Eii[e] += (w * Jz)[None] * Ji.T
                Eij[e] += (w * Jz)[None] * Jj.T
        # pose x pose block (update_lhs / update_rhs drop negative indices)
        A = np.zeros((Pn * 6, Pn * 6))
        b = np.zeros(Pn * 6)
        for blk, (ra, rb) in enumerate([(ii, ii), (ii, jj), (jj, ii), (jj, jj)]):
            for e in range(E):
                a_, b_ = int(ra[e]) - t0, int(rb[e]) - t0
                if a_ >= 0 and b_ >= 0:
                    A[6 * a_:6 * a_ + 6, 6 * b_:6 * b_ + 6] += Hs[blk, e]
        for blk, ra in enumerate([ii, jj]):
            for e in range(E):
                a_ = int(ra[e]) - t0
                if a_ >= 0:
                    b[6 * a_:6 * a_ + 6] += vs[blk, e]

        def solve(Am, bm):
            L = Am.copy()
            dg = np.diag(L).copy()
            L[np.diag_indices_from(L)] = dg + ep + lm * dg
            try:
                c = np.linalg.cholesky(L)
            except np.linalg.LinAlgError:
                return np.zeros((Pn, 6))
            return np.linalg.solve(c.T, np.linalg.solve(c, bm)).reshape(Pn, 6)

        if motion_only:
            dx = solve(A, b)
        else:
            alpha = 0.05
            m = (sens[kx] > 0).astype(np.float64).reshape(K, Pp)
            C = np.zeros((K, Pp))
            w = np.zeros((K, Pp))
            np.add.at(C, kk_exp[Pn:], Cii)
            np.add.at(w, kk_exp[Pn:], bz)
            C = C + m * alpha + (1 - m) * eta
            w = w - m * alpha * (disps[kx] - sens[kx]).reshape(K, Pp)
            Q = 1.0 / C
            Ei = np.zeros((Pn, 6, Pp))
            for e in range(E):
                if t0 <= ii[e] < t1:
                    Ei[ii[e] - t0] += Eii[e]
            Eall = np.concatenate([Ei, Eij], 0)  # block n: pose jj_exp[n], disparity frame kk_exp[n]
            S = np.zeros((Pn * 6, Pn * 6))
            vS = np.zeros(Pn * 6)
            blocks = [n for n in range(len(jj_exp)) if t0 <= jj_exp[n] < t1]
            for na in blocks:
                pa = jj_exp[na] - t0
                vS[6 * pa:6 * pa + 6] += Eall[na] @ (Q[kk_exp[na]] * w[kk_exp[na]])
                for nb in blocks:
                    if kk_exp[na] == kk_exp[nb]:
                        pb = jj_exp[nb] - t0
                        S[6 * pa:6 * pa + 6, 6 * pb:6 * pb + 6] += (Eall[na] * Q[kk_exp[na]][None]) @ Eall[nb].T
            dx = solve(A - S, b - vS)
            dw = np.zeros((K, Pp))
            for n in range(len(jj_exp)):
                ix = jj_exp[n] - t0
                if ix <= 0 or ix >= Pn:
                    continue  # EvT6x1_kernel:1085
                dw[kk_exp[n]] += dx[ix] @ Eall[n]
            dz = Q * (w - dw)
            disps[kx] += dz.reshape(K, ht, wd)
        poses[t0:t1] = se3.se3_retr(poses[t0:t1], dx)
    return poses, disps, dx, dz
