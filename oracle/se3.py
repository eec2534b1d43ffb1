"""SO3 / SE3 closed forms, numpy, dtype-generic (float32 or float64).

ORACLE (test infrastructure). Restates csrc/lietorch_ext/so3.h and se3.h:
data layout [tx,ty,tz,qx,qy,qz,qw]; tangent [tau(3), phi(3)].
Quaternions are re-normalised on load and after every product
(so3.h:36-38), EPS = 1e-6 (common.h:13).
"""

import numpy as np

EPS = 1e-6


def _f(x, dtype=None):
    x = np.asarray(x)
    if dtype is not None:
        x = x.astype(dtype)
    return x


def quat_normalize(q):
    """so3.h:36-38 (constructor normalises)."""
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def quat_mul(a, b):
    """Hamilton product, (x,y,z,w) layout; Eigen Quaternion operator*."""
    ax, ay, az, aw = np.moveaxis(a, -1, 0)
    bx, by, bz, bw = np.moveaxis(b, -1, 0)
    return np.stack(
        [
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by + ay * bw + az * bx - ax * bz,
            aw * bz + az * bw + ax * by - ay * bx,
            aw * bw - ax * bx - ay * by - az * bz,
        ],
        axis=-1,
    )


def quat_conj(q):
    return q * np.array([-1, -1, -1, 1], dtype=q.dtype)


def so3_mul(a, b):
    """so3.h:46-48."""
    return quat_normalize(quat_mul(quat_normalize(a), quat_normalize(b)))


def so3_inv(q):
    """so3.h:42."""
    return quat_normalize(quat_conj(quat_normalize(q)))


def so3_act(q, p):
    """so3.h:50-55: p + w*uv + qv x uv with uv = 2 (qv x p)."""
    q = quat_normalize(q)
    qv, w = q[..., :3], q[..., 3:4]
    uv = np.cross(qv, p)
    uv = uv + uv
    return p + w * uv + np.cross(qv, uv)


def so3_matrix(q):
    """Eigen toRotationMatrix (so3.h:63-65)."""
    q = quat_normalize(q)
    x, y, z, w = np.moveaxis(q, -1, 0)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = np.stack(
        [
            1 - (tyy + tzz), txy - twz, txz + twy,
            txy + twz, 1 - (txx + tzz), tyz - twx,
            txz - twy, tyz + twx, 1 - (txx + tyy),
        ],
        axis=-1,
    )
    return R.reshape(q.shape[:-1] + (3, 3))


def hat(phi):
    """so3.h:88-93."""
    o = np.zeros_like(phi[..., 0])
    x, y, z = np.moveaxis(phi, -1, 0)
    return np.stack([o, -z, y, z, o, -x, -y, x, o], axis=-1).reshape(phi.shape[:-1] + (3, 3))


def so3_exp(phi):
    """so3.h:133-151."""
    dt = phi.dtype
    theta2 = np.sum(phi * phi, axis=-1)
    theta = np.sqrt(theta2)
    theta4 = theta2 * theta2
    small = theta < EPS
    th = np.where(small, np.ones_like(theta), theta)
    imag = np.where(small, dt.type(0.5) - dt.type(1.0 / 48.0) * theta2 + dt.type(1.0 / 3840.0) * theta4,
                    np.sin(dt.type(0.5) * th) / th)
    real = np.where(small, dt.type(1) - dt.type(1.0 / 8.0) * theta2 + dt.type(1.0 / 384.0) * theta4,
                    np.cos(dt.type(0.5) * th))
    q = np.concatenate([imag[..., None] * phi, real[..., None]], axis=-1).astype(dt)
    return quat_normalize(q)


def so3_log(q):
    """so3.h:96-131 (atan-based)."""
    q = quat_normalize(q)
    dt = q.dtype
    v, w = q[..., :3], q[..., 3]
    sn = np.sum(v * v, axis=-1)
    n = np.sqrt(sn)
    with np.errstate(divide="ignore", invalid="ignore"):
        small = dt.type(2) / w - dt.type(2.0 / 3.0) * sn / (w * w * w)
        nn = np.where(n > 0, n, np.ones_like(n))
        pi_branch = np.where(w > 0, dt.type(np.pi) / nn, -dt.type(np.pi) / nn)
        atan_branch = dt.type(2) * np.arctan(nn / np.where(np.abs(w) < EPS, np.ones_like(w), w)) / nn
    f = np.where(sn < EPS * EPS, small, np.where(np.abs(w) < EPS, pi_branch, atan_branch))
    return (f[..., None] * v).astype(dt)


def so3_left_jacobian(phi):
    """so3.h:153-168."""
    dt = phi.dtype
    Phi = hat(phi)
    Phi2 = Phi @ Phi
    theta2 = np.sum(phi * phi, axis=-1)
    theta = np.sqrt(theta2)
    small = theta < EPS
    t2 = np.where(small, np.ones_like(theta2), theta2)
    th = np.where(small, np.ones_like(theta), theta)
    c1 = np.where(small, dt.type(0.5) - dt.type(1.0 / 24.0) * theta2, (1.0 - np.cos(th)) / t2)
    c2 = np.where(small, dt.type(1.0 / 6.0) - dt.type(1.0 / 120.0) * theta2, (th - np.sin(th)) / (t2 * th))
    I = np.eye(3, dtype=dt)
    return (I + c1[..., None, None] * Phi + c2[..., None, None] * Phi2).astype(dt)


def so3_left_jacobian_inverse(phi):
    """so3.h:170-184."""
    dt = phi.dtype
    Phi = hat(phi)
    Phi2 = Phi @ Phi
    theta2 = np.sum(phi * phi, axis=-1)
    theta = np.sqrt(theta2)
    small = theta < EPS
    th = np.where(small, np.ones_like(theta), theta)
    half = dt.type(0.5) * th
    c2 = np.where(small, dt.type(1.0 / 12.0),
                  (dt.type(1) - th * np.cos(half) / (dt.type(2) * np.sin(half))) / (th * th))
    I = np.eye(3, dtype=dt)
    return (I + dt.type(-0.5) * Phi + c2[..., None, None] * Phi2).astype(dt)


# ----------------------------------------------------------------------------- SE3


def se3_split(X):
    return X[..., :3], X[..., 3:7]


def se3_mul(A, B):
    """se3.h:48-50."""
    ta, qa = se3_split(A)
    tb, qb = se3_split(B)
    q = so3_mul(qa, qb)
    t = ta + so3_act(qa, tb)
    return np.concatenate([t, q], axis=-1)


def se3_inv(X):
    """se3.h:40."""
    t, q = se3_split(X)
    qi = so3_inv(q)
    return np.concatenate([-so3_act(qi, t), qi], axis=-1)


def se3_act4(X, p):
    """se3.h:54-58: [R p_xyz + t p_w, p_w]."""
    t, q = se3_split(X)
    xyz = so3_act(q, p[..., :3]) + t * p[..., 3:4]
    return np.concatenate([xyz, p[..., 3:4]], axis=-1)


def se3_act3(X, p):
    """se3.h:52."""
    t, q = se3_split(X)
    return so3_act(q, p) + t


def se3_adj_matrix(X):
    """se3.h:60-69: [[R, t^ R],[0, R]]."""
    t, q = se3_split(X)
    R = so3_matrix(q)
    tx = hat(t)
    Z = np.zeros_like(R)
    top = np.concatenate([R, tx @ R], axis=-1)
    bot = np.concatenate([Z, R], axis=-1)
    return np.concatenate([top, bot], axis=-2)


def se3_adj(X, a):
    """se3.h:81: Adj(X) a."""
    return np.einsum("...ij,...j->...i", se3_adj_matrix(X), a)


def se3_adjT(X, a):
    """se3.h:83: Adj(X)^T a."""
    return np.einsum("...ji,...j->...i", se3_adj_matrix(X), a)


def se3_exp(xi):
    """se3.h:127-136."""
    tau, phi = xi[..., :3], xi[..., 3:]
    q = so3_exp(phi)
    t = np.einsum("...ij,...j->...i", so3_left_jacobian(phi), tau)
    return np.concatenate([t, q], axis=-1).astype(xi.dtype)


def se3_log(X):
    """se3.h:117-125."""
    t, q = se3_split(X)
    phi = so3_log(q)
    Vinv = so3_left_jacobian_inverse(phi)
    tau = np.einsum("...ij,...j->...i", Vinv, t)
    return np.concatenate([tau, phi], axis=-1).astype(X.dtype)


def se3_retr(X, a):
    """groups.py:147-150: Exp(a) * X."""
    return se3_mul(se3_exp(a), X)


def se3_matrix(X):
    """se3.h:71-76."""
    t, q = se3_split(X)
    R = so3_matrix(q)
    T = np.zeros(X.shape[:-1] + (4, 4), dtype=X.dtype)
    T[..., :3, :3] = R
    T[..., :3, 3] = t
    T[..., 3, 3] = 1
    return T


def se3_identity(n, dtype=np.float32):
    X = np.zeros((n, 7), dtype=dtype)
    X[:, 6] = 1
    return X
