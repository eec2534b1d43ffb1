"""Projective geometry of the dense BA: camera models, reprojection, Jacobians.

ORACLE (test infrastructure). numpy, dtype-generic. Follows
vipe/utils/cameras.py:123-354 (pinhole + MEI) and
vipe/slam/maths/geom.py:91-298 (actp, iproj_i_proj_j_disp).
"""

import numpy as np

from . import se3

MIN_DEPTH = 0.1  # cameras.py:48


def expand_edge_multiview(ii, jj, n_views, cross_view_idx=None, cross=True, view_offset=0):
    """buffer.py:318-361. Returns pi, qi, di, pj, qj, dj (each [M*V])."""
    ii = np.asarray(ii, dtype=np.int64)
    jj = np.asarray(jj, dtype=np.int64)
    V = n_views
    qi = np.tile(np.arange(V, dtype=np.int64)[None], (len(ii), 1))
    pi = np.tile(ii[:, None], (1, V))
    qj = np.tile(np.arange(V, dtype=np.int64)[None], (len(jj), 1))
    pj = np.tile(jj[:, None], (1, V))
    if cross:
        m = ii == jj
        if m.any():
            if cross_view_idx is None:
                cross_view_idx = [(i + 1) % V for i in range(V)]
            cv = np.asarray(cross_view_idx, dtype=np.int64)
            # cross_view_idx[t, v] = (t, cv[v])  (buffer.py:174-176)
            qj[m] = cv[qi[m]]
    qj = (qj + view_offset) % V
    di = pi * V + qi
    dj = pj * V + qj
    return tuple(a.reshape(-1) for a in (pi, qi, di, pj, qj, dj))


def pixel_grid(ht, wd, dtype):
    """geom.py:48-55: u = 0..wd-1, v = 0..ht-1."""
    v, u = np.meshgrid(np.arange(ht, dtype=dtype), np.arange(wd, dtype=dtype), indexing="ij")
    return u, v


def iproj_disp(disps, intr, model, compute_jf=False):
    """cameras.py:131-159 (pinhole), :228-281 (mei). disps [M,ht,wd]; intr [M,4|5].

    Returns X0 [M,ht,wd,4] and Jf [M,ht,wd,4,1+D] (d X0 / d(focal, k1)) or None.
    """
    dt = disps.dtype
    M, ht, wd = disps.shape
    u, v = pixel_grid(ht, wd, dt)
    I = intr.reshape(M, 1, 1, -1)
    fx, fy, cx, cy = I[..., 0], I[..., 1], I[..., 2], I[..., 3]
    one = np.ones_like(disps)
    Jf = None
    if model == "pinhole":
        X = (u - cx) / fx
        Y = (v - cy) / fy
        if compute_jf:
            Jf = np.zeros(disps.shape + (4, 1), dtype=dt)
            Jf[..., 0, 0] = -X / fx
            Jf[..., 1, 0] = -Y / fy
    elif model == "mei":
        k1 = I[..., 4]
        ub = (u - cx) / fx
        vb = (v - cy) / fy
        r2 = ub**2 + vb**2
        q = np.sqrt(1 + (1 - k1**2) * r2)
        factor = (k1 + q) / (1 + r2)
        X = ub * factor / (factor - k1)
        Y = vb * factor / (factor - k1)
        if compute_jf:
            Jf = np.zeros(disps.shape + (4, 2), dtype=dt)
            f_num = (-(k1**3) * r2**2 - k1**3 * r2 - k1**2 * q * r2 - k1 * q**2 * r2 - k1 * q**2
                     + k1 * r2**2 + k1 * r2 - q**3)
            f_den = fx * q * (k1**2 * r2**2 - 2 * k1 * q * r2 + q**2)
            Jf[..., 0, 0] = ub * f_num / f_den
            Jf[..., 1, 0] = vb * f_num / f_den
            k_num = (k1 + q) * (k1 * r2 + q * (r2 + 1) - q) - (k1 * r2 - q) * (-k1 * (r2 + 1) + k1 + q)
            k_den = q * (-k1 * (r2 + 1) + k1 + q) ** 2
            Jf[..., 0, 1] = ub * k_num / k_den
            Jf[..., 1, 1] = vb * k_num / k_den
    else:
        raise ValueError(model)
    X0 = np.stack([X, Y, one, disps], axis=-1).astype(dt)
    return X0, Jf


def proj_points(X1, intr, model, jac=False, compute_jf=False):
    """cameras.py:161-207 (pinhole), :283-336 (mei). X1 [M,ht,wd,4]; intr [M,4|5]."""
    dt = X1.dtype
    M = X1.shape[0]
    I = intr.reshape(M, 1, 1, -1)
    fx, fy, cx, cy = I[..., 0], I[..., 1], I[..., 2], I[..., 3]
    X, Y, Z = X1[..., 0], X1[..., 1], X1[..., 2]
    Z = np.where(Z < dt.type(MIN_DEPTH), np.ones_like(Z), Z)  # cameras.py:175-177
    Jp = Jf = None
    if model == "pinhole":
        d = 1 / Z
        x = fx * (X * d) + cx
        y = fy * (Y * d) + cy
        if jac:
            o = np.zeros_like(d)
            Jp = np.stack([fx * d, o, -fx * X * d * d, o, o, fy * d, -fy * Y * d * d, o], axis=-1)
            Jp = Jp.reshape(X.shape + (2, 4))
        if compute_jf:
            Jf = np.zeros(X.shape + (2, 1), dtype=dt)
            Jf[..., 0, 0] = X * d
            Jf[..., 1, 0] = Y * d
    elif model == "mei":
        k1 = I[..., 4]
        r = np.sqrt(X**2 + Y**2 + Z**2)
        rbase = Z + k1 * r
        d = 1 / rbase
        x = fx * (X * d) + cx
        y = fy * (Y * d) + cy
        if jac:
            o = np.zeros_like(d)
            rd = rbase**2 * r
            Jp = np.stack(
                [
                    fx * (-k1 * X**2 + rbase * r) / rd, -fx * k1 * X * Y / rd, -fx * X * (k1 * Z + r) / rd, o,
                    -fy * k1 * X * Y / rd, fy * (-k1 * Y**2 + rbase * r) / rd, -fy * Y * (k1 * Z + r) / rd, o,
                ],
                axis=-1,
            ).reshape(X.shape + (2, 4))
        if compute_jf:
            Jf = np.zeros(X.shape + (2, 2), dtype=dt)
            Jf[..., 0, 0] = X * d
            Jf[..., 1, 0] = Y * d
            Jf[..., 0, 1] = -fx * r * X * d**2
            Jf[..., 1, 1] = -fy * r * Y * d**2
    else:
        raise ValueError(model)
    coords = np.stack([x, y], axis=-1).astype(dt)
    return coords, Jp, Jf


def scaled_intrinsics(intr, scale, model):
    """cameras.py:212-213, :345-348."""
    out = np.array(intr, copy=True)
    out[..., :4] = out[..., :4] * out.dtype.type(scale)
    return out


def pinhole_of(intr, model):
    """cameras.py:209-210, :338-343."""
    if model == "pinhole":
        return intr
    out = np.array(intr[..., :4], copy=True)
    out[..., 0:2] = out[..., 0:2] / (1 + intr[..., 4:5])
    return out


def reproject(poses, disps, intr, rig, pi, pj, qi, qj, di, model="pinhole", jacobian=False, jacobian_f=False):
    """geom.py:187-298. poses [N,7], disps [NV,ht,wd], intr [Q,4+D] already at 1/8 scale, rig [Q,7].

    Returns coords [M,ht,wd,2], valid [M,ht,wd], and if jacobian:
    Ji, Jj [M,ht,wd,2,6], Jz [M,ht,wd,2]; if jacobian_f: Jfi, Jfj [M,ht,wd,2,1+D].
    """
    dt = disps.dtype
    poses = poses.astype(dt)
    intr = intr.astype(dt)
    rig = rig.astype(dt)
    d = disps[di]
    X0, Jf0 = iproj_disp(d, intr[qi], model, compute_jf=jacobian_f)

    Gij = se3.se3_mul(poses[pj], se3.se3_inv(poses[pi]))  # geom.py:251
    Rji = se3.se3_inv(rig[qj])
    T = se3.se3_mul(se3.se3_mul(Rji, Gij), rig[qi])  # geom.py:252
    X1 = se3.se3_act4(T[:, None, None, :], X0)  # geom.py:106

    coords, Jp, Jfj = proj_points(X1, intr[qj], model, jac=jacobian, compute_jf=jacobian_f)
    valid = ((X1[..., 2] > dt.type(MIN_DEPTH)) & (X0[..., 2] > dt.type(MIN_DEPTH))).astype(dt)  # geom.py:263
    out = {"coords": coords, "valid": valid}
    if not jacobian:
        return out

    X, Y, Z, dd = X1[..., 0], X1[..., 1], X1[..., 2], X1[..., 3]
    o = np.zeros_like(dd)
    Ja = np.stack(  # geom.py:114-145
        [dd, o, o, o, Z, -Y, o, dd, o, -Z, o, X, o, o, dd, Y, -X, o, o, o, o, o, o, o], axis=-1
    ).reshape(X.shape + (4, 6))
    # Ja rows <- Adj(R_qj^-1)^T row  (geom.py:273; adjT(a) = Adj^T a, se3.h:83)
    A_r = se3.se3_adj_matrix(Rji)  # [M,6,6]
    Ja = np.einsum("mij,mhwri->mhwrj", A_r, Ja)
    Jj = np.einsum("mhwcr,mhwrk->mhwck", Jp, Ja)  # geom.py:275
    A_g = se3.se3_adj_matrix(Gij)
    Ji = -np.einsum("mij,mhwci->mhwcj", A_g, Jj)  # geom.py:277
    # Jz = Jp . (T * [0,0,0,1]) (geom.py:280-281)
    e4 = np.zeros((1, 1, 1, 4), dtype=dt)
    e4[..., 3] = 1
    TJz = se3.se3_act4(T[:, None, None, :], np.broadcast_to(e4, X0.shape))
    Jz = np.einsum("mhwcr,mhwr->mhwc", Jp, TJz)
    out.update({"Ji": Ji.astype(dt), "Jj": Jj.astype(dt), "Jz": Jz.astype(dt)})
    if jacobian_f:
        # Jfi = Jp . (T * dX0/df) per column  (geom.py:286-288)
        cols = []
        for k in range(Jf0.shape[-1]):
            cols.append(np.einsum("mhwcr,mhwr->mhwc", Jp, se3.se3_act4(T[:, None, None, :], Jf0[..., k])))
        out["Jfi"] = np.stack(cols, axis=-1).astype(dt)
        out["Jfj"] = Jfj.astype(dt)
    return out
