"""frame_distance, depth_filter, projmap, iproj - numpy restatements of the live / bound slam_ext kernels.

ORACLE (test infrastructure). Follows csrc/slam_ext/geom_kernels.cu:106-156 (actSO3/actSE3/relSE3, raw quaternions,
no re-normalisation), :434-519 (projmap), :521-676 (frame_distance), :678-793 (depth_filter), :795-861 (iproj);
MIN_DEPTH = 0.25 (:33).  No CPU path exists in the reference: "parity unpinned" by any reference output, checked
against the independent oracle.geom reprojection in the tests.
"""

import numpy as np

MIN_DEPTH = np.float32(0.25)


def _act_so3(q, X):
    uv = np.stack([2 * (q[..., 1] * X[..., 2] - q[..., 2] * X[..., 1]), 2 * (q[..., 2] * X[..., 0] - q[..., 0] * X[..., 2]),
                   2 * (q[..., 0] * X[..., 1] - q[..., 1] * X[..., 0])], -1)
    return np.stack([X[..., 0] + q[..., 3] * uv[..., 0] + (q[..., 1] * uv[..., 2] - q[..., 2] * uv[..., 1]),
                     X[..., 1] + q[..., 3] * uv[..., 1] + (q[..., 2] * uv[..., 0] - q[..., 0] * uv[..., 2]),
                     X[..., 2] + q[..., 3] * uv[..., 2] + (q[..., 0] * uv[..., 1] - q[..., 1] * uv[..., 0])], -1)


def rel_se3(pi, pj):
    ti, qi, tj, qj = pi[..., :3], pi[..., 3:], pj[..., :3], pj[..., 3:]
    q = np.stack([-qj[..., 3] * qi[..., 0] + qj[..., 0] * qi[..., 3] - qj[..., 1] * qi[..., 2] + qj[..., 2] * qi[..., 1],
                  -qj[..., 3] * qi[..., 1] + qj[..., 1] * qi[..., 3] - qj[..., 2] * qi[..., 0] + qj[..., 0] * qi[..., 2],
                  -qj[..., 3] * qi[..., 2] + qj[..., 2] * qi[..., 3] - qj[..., 0] * qi[..., 1] + qj[..., 1] * qi[..., 0],
                  qj[..., 3] * qi[..., 3] + qj[..., 0] * qi[..., 0] + qj[..., 1] * qi[..., 1] + qj[..., 2] * qi[..., 2]], -1)
    return tj - _act_so3(q, ti), q


def _grid(ht, wd):
    v, u = np.meshgrid(np.arange(ht, dtype=np.float32), np.arange(wd, dtype=np.float32), indexing="ij")
    return u, v


def frame_distance(poses, disps, intrinsics, pi, pj, qi, qj, di, beta):
    """geom_kernels.cu:521-676."""
    poses, disps, intrinsics = (np.asarray(a, np.float32) for a in (poses, disps, intrinsics))
    beta = np.float32(beta)
    _, ht, wd = disps.shape
    u, v = _grid(ht, wd)
    out = np.zeros(len(pi), np.float32)
    for b in range(len(pi)):
        t, q = rel_se3(poses[pi[b]], poses[pj[b]])
        fxi, fyi, cxi, cyi = intrinsics[qi[b]]
        fxj, fyj, cxj, cyj = intrinsics[qj[b]]
        d = disps[di[b]]
        Xi = np.stack([(u - cxi) / fxi, (v - cyi) / fyi, np.ones_like(u)], -1)
        Xj = _act_so3(q, Xi) + d[..., None] * t
        dist = np.sqrt((fxj * (Xj[..., 0] / Xj[..., 2]) + cxj - u) ** 2 + (fyj * (Xj[..., 1] / Xj[..., 2]) + cyj - v) ** 2)
        ok = Xj[..., 2] > MIN_DEPTH
        accum = beta * np.sum(dist[ok], dtype=np.float64)
        valid = beta * ok.sum()
        Xt = Xi + d[..., None] * t
        dist = np.sqrt((fxj * (Xt[..., 0] / Xt[..., 2]) + cxj - u) ** 2 + (fyj * (Xt[..., 1] / Xt[..., 2]) + cyj - v) ** 2)
        ok = Xt[..., 2] > MIN_DEPTH
        accum += (1 - beta) * np.sum(dist[ok], dtype=np.float64)
        valid += (1 - beta) * ok.sum()
        total = float(ht * wd)
        out[b] = 1000.0 if valid / (total + 1e-8) < 0.75 else accum / valid
    return out


def depth_filter(poses, disps, intrinsics, inds, thresh):
    """geom_kernels.cu:678-793 -> counter [num,ht,wd] (integer valued float32)."""
    poses, disps, intr = (np.asarray(a, np.float32) for a in (poses, disps, intrinsics))
    n, ht, wd = disps.shape
    fx, fy, cx, cy = intr
    u, v = _grid(ht, wd)
    out = np.zeros((len(inds), ht, wd), np.float32)
    for b, ix in enumerate(inds):
        for nb in range(6):
            jx = ix - nb - 1 if nb < 3 else ix + nb - 2
            if jx < 0 or jx >= n:
                continue
            t, q = rel_se3(poses[ix], poses[jx])
            Xi = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
            Xj = _act_so3(q, Xi) + disps[ix][..., None] * t
            uj = fx * (Xj[..., 0] / Xj[..., 2]) + cx
            vj = fy * (Xj[..., 1] / Xj[..., 2]) + cy
            dj = disps[ix] / Xj[..., 2]
            u0 = np.floor(uj).astype(np.int64)
            v0 = np.floor(vj).astype(np.int64)
            ok = (u0 >= 0) & (v0 >= 0) & (u0 < wd - 1) & (v0 < ht - 1)
            u0c, v0c = np.clip(u0, 0, wd - 2), np.clip(v0, 0, ht - 2)
            hit = np.zeros((ht, wd), bool)
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / dj.astype(np.float64)
                for (a, c) in ((0, 0), (0, 1), (1, 0), (1, 1)):
                    dn = disps[jx][v0c + a, u0c + c].astype(np.float64)
                    hit |= np.abs(inv - 1.0 / dn) < float(np.float32(thresh[b]))
            out[b] += (ok & hit).astype(np.float32)
    return out


def projmap(poses, disps, intrinsics, ii, jj):
    poses, disps, intr = (np.asarray(a, np.float32) for a in (poses, disps, intrinsics))
    _, ht, wd = disps.shape
    fx, fy, cx, cy = intr
    u, v = _grid(ht, wd)
    coords = np.zeros((len(ii), ht, wd, 3), np.float32)
    valid = np.zeros((len(ii), ht, wd, 1), np.float32)
    for b in range(len(ii)):
        t, q = rel_se3(poses[ii[b]], poses[jj[b]])
        Xi = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
        Xj = _act_so3(q, Xi) + disps[ii[b]][..., None] * t
        ok = Xj[..., 2] > np.float32(0.01)
        with np.errstate(divide="ignore", invalid="ignore"):
            coords[b, ..., 0] = np.where(ok, fx * (Xj[..., 0] / Xj[..., 2]) + cx, u)
            coords[b, ..., 1] = np.where(ok, fy * (Xj[..., 1] / Xj[..., 2]) + cy, v)
        valid[b, ..., 0] = Xj[..., 2] > MIN_DEPTH
    return coords, valid


def iproj(poses, disps, intrinsics):
    poses, disps, intr = (np.asarray(a, np.float32) for a in (poses, disps, intrinsics))
    n, ht, wd = disps.shape
    fx, fy, cx, cy = intr
    u, v = _grid(ht, wd)
    out = np.zeros((n, ht, wd, 3), np.float32)
    for b in range(n):
        Xi = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
        X = _act_so3(poses[b, 3:], Xi) + disps[b][..., None] * poses[b, :3]
        out[b] = X / disps[b][..., None]
    return out
