"""PyTorch-CPU corr + BA path: the `cpu_baseline` leg of bench.py.

ORACLE (test infrastructure) - imported only by tests/ and bench.py's cpu_baseline; nothing under vipe_amd/ uses it.
A vectorised torch-CPU (fp32) restatement of the non-GRU part of one update iteration on a mono pinhole graph:
  a1  all-pairs correlation volume + 3x avg-pool pyramid      droid_net.py:56-69,94-102   (matmul + avg_pool2d)
  a2  4-level 7x7 bilinear lookup                              droid_net.py:71-82, correlation_kernels.cu:22-66
  a5-a10  dense BA on SE3 (+) inverse depth, Schur + solve     buffer.py:373-525, terms.py:94-303, solver.py:117-197
Same arithmetic as oracle/corr.py + oracle/ba.py (tests/test_torch_cpu_path.py checks it against them), organised for
BLAS / OpenMP: per-edge Jacobian blocks by einsum, the Schur complement as one Gram matrix per source frame.
"""

import torch
import torch.nn.functional as F

MIN_DEPTH = 0.1  # cameras.py:48


# ---------------------------------------------------------------- correlation
def corr_pyramid(fmap1, fmap2, num_levels=4):
    """fmap [E,C,h,w] fp32 -> list of [E,h,w,h>>i,w>>i]."""
    E, C, h, w = fmap1.shape
    vol = torch.matmul((fmap1.reshape(E, C, h * w) / 4.0).transpose(1, 2), fmap2.reshape(E, C, h * w) / 4.0)
    vol = vol.reshape(E * h * w, 1, h, w)
    pyr = []
    for i in range(num_levels):
        pyr.append(vol.view(E, h, w, h >> i, w >> i))
        if i + 1 < num_levels:
            vol = F.avg_pool2d(vol, 2, stride=2)
    return pyr


def corr_lookup(pyr, coords, radius=3):
    """coords [E,h,w,2] (x,y) -> [E, L*(2r+1)^2, h, w]; channel = level*(49) + ix*(2r+1) + iy (x-offset major, the
    kernel's [i][j] order).  Zero outside the level, weights as correlation_kernels.cu:56-62."""
    E, h, w, _ = coords.shape
    r, rd = radius, 2 * radius + 1
    outs = []
    for lvl, vol in enumerate(pyr):
        h2, w2 = vol.shape[-2:]
        c = coords / float(2 ** lvl)
        x, y = c[..., 0].reshape(-1), c[..., 1].reshape(-1)
        fx, fy = torch.floor(x), torch.floor(y)
        dx, dy = (x - fx)[:, None, None], (y - fy)[:, None, None]
        ix, iy = fx.long(), fy.long()
        off = torch.arange(rd + 1) - r
        xs = ix[:, None] + off[None]  # [EP, 8]
        ys = iy[:, None] + off[None]
        inb = ((xs >= 0) & (xs < w2))[:, :, None] & ((ys >= 0) & (ys < h2))[:, None, :]  # [EP, i, j]
        idx = ys.clamp(0, h2 - 1)[:, None, :] * w2 + xs.clamp(0, w2 - 1)[:, :, None]
        s = torch.gather(vol.reshape(E * h * w, h2 * w2), 1, idx.reshape(E * h * w, -1)).view(-1, rd + 1, rd + 1)
        s = s * inb
        # out[i][j] = s[i][j] (1-dx)(1-dy) + s[i][j+1] (1-dx) dy + s[i+1][j] dx (1-dy) + s[i+1][j+1] dx dy
        o = (s[:, :-1, :-1] * ((1 - dx) * (1 - dy)) + s[:, :-1, 1:] * ((1 - dx) * dy)
             + s[:, 1:, :-1] * (dx * (1 - dy)) + s[:, 1:, 1:] * (dx * dy))
        outs.append(o.reshape(E, h, w, rd * rd).permute(0, 3, 1, 2))
    return torch.cat(outs, 1)


# ---------------------------------------------------------------- SE3 (se3.h, so3.h), rows [t, q(x,y,z,w)]
def _qmul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                        aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], -1)


def _qrot(q, p):
    qv, w = q[..., :3], q[..., 3:4]
    uv = 2 * torch.linalg.cross(qv.expand_as(p), p)
    return p + w * uv + torch.linalg.cross(qv.expand_as(p), uv)


def _norm(q):
    return q / q.norm(dim=-1, keepdim=True)


def se3_mul(A, B):
    qa, qb = _norm(A[..., 3:]), _norm(B[..., 3:])
    return torch.cat([A[..., :3] + _qrot(qa, B[..., :3]), _norm(_qmul(qa, qb))], -1)


def se3_inv(A):
    qi = _norm(A[..., 3:]) * torch.tensor([-1.0, -1.0, -1.0, 1.0], dtype=A.dtype)
    return torch.cat([-_qrot(qi, A[..., :3]), qi], -1)


def _hat(v):
    o = torch.zeros_like(v[..., 0])
    return torch.stack([o, -v[..., 2], v[..., 1], v[..., 2], o, -v[..., 0], -v[..., 1], v[..., 0], o], -1).reshape(
        v.shape[:-1] + (3, 3))


def _rotmat(q):
    eye = torch.eye(3, dtype=q.dtype).expand(q.shape[:-1] + (3, 3))
    return torch.stack([_qrot(q, eye[..., :, k]) for k in range(3)], -1)


def se3_adj(A):
    """Adj = [[R, t^ R], [0, R]] (se3.h:60-69) -> [...,6,6]"""
    R = _rotmat(_norm(A[..., 3:]))
    tR = _hat(A[..., :3]) @ R
    top = torch.cat([R, tR], -1)
    bot = torch.cat([torch.zeros_like(R), R], -1)
    return torch.cat([top, bot], -2)


def se3_exp(xi):
    """se3.h:119-137, so3.h:139-184 (Taylor below theta < 1e-6 is irrelevant at fp32 step sizes but kept)."""
    tau, phi = xi[..., :3], xi[..., 3:]
    th2 = (phi * phi).sum(-1, keepdim=True)
    th = th2.sqrt()
    small = th < 1e-6
    ths = torch.where(small, torch.ones_like(th), th)
    imag = torch.where(small, 0.5 - th2 / 48.0, torch.sin(0.5 * ths) / ths)
    real = torch.where(small, 1.0 - th2 / 8.0, torch.cos(0.5 * ths))
    q = torch.cat([imag * phi, real], -1)
    a = torch.where(small, torch.full_like(th, 0.5), (1 - torch.cos(ths)) / (ths * ths))
    b = torch.where(small, torch.full_like(th, 1.0 / 6.0), (ths - torch.sin(ths)) / (ths * ths * ths))
    Phi = _hat(phi)
    V = torch.eye(3, dtype=xi.dtype) + a[..., None] * Phi + b[..., None] * (Phi @ Phi)
    return torch.cat([(V @ tau[..., None])[..., 0], _norm(q)], -1)


# ---------------------------------------------------------------- dense BA (mono pinhole, pose + disparity)
def _linearise(poses, disps, intr8, ii, jj, ht, wd):
    """geom.py:187-298 for V = 1, pinhole: coords, valid, Ji, Jj, Jz per edge ([E,P,...])."""
    E = ii.shape[0]
    fx, fy, cx, cy = intr8.unbind(-1)
    v, u = torch.meshgrid(torch.arange(ht, dtype=poses.dtype), torch.arange(wd, dtype=poses.dtype), indexing="ij")
    X0 = torch.stack([(u - cx) / fx, (v - cy) / fy, torch.ones_like(u)], -1).reshape(1, -1, 3)
    d = disps[ii].reshape(E, -1)
    G = se3_mul(poses[jj], se3_inv(poses[ii]))
    q, t = G[:, None, 3:], G[:, None, :3]
    X1 = _qrot(q, X0.expand(E, -1, 3)) + t * d[..., None]
    X, Y, Zr = X1.unbind(-1)
    Z = torch.where(Zr < MIN_DEPTH, torch.ones_like(Zr), Zr)
    s = 1.0 / Z
    coords = torch.stack([fx * X * s + cx, fy * Y * s + cy], -1)
    valid = (Zr > MIN_DEPTH).to(poses.dtype)
    o = torch.zeros_like(s)
    Jp = torch.stack([fx * s, o, -fx * X * s * s, o, fy * s, -fy * Y * s * s], -1).reshape(E, -1, 2, 3)
    # Ja = d(T X0)/d xi with the TRANSFORMED point (geom.py:114-145); the clamped Z enters only through Jp
    Ja = torch.stack([d, o, o, o, Zr, -Y, o, d, o, -Zr, o, X, o, o, d, Y, -X, o], -1).reshape(E, -1, 3, 6)
    Jj = Jp @ Ja
    Ji = -(Jj @ se3_adj(G)[:, None])  # row vectors: (Adj^T J^T)^T = J Adj
    Jz = (Jp @ t[..., None].expand(E, Jp.shape[1], 3, 1))[..., 0]
    return coords, valid, Ji, Jj, Jz


def bundle_adjustment(poses, disps, disps_sens, intrinsics, target, weight, eta, ii, jj, t0, t1, n_iters,
                      pose_damping, pose_ep, alpha=0.001, weight_scale=0.001):
    """buffer.py:373-525 for the bench configuration: V = 1, pinhole, poses in [t0,t1) free, every source frame's
    disparity free, sensor-depth prior on frames whose disps_sens sums > 0.  poses [N,7], disps/disps_sens/eta
    [N,ht,wd], target/weight [E,P,2], ii/jj [E] int64 -> (poses, disps) updated copies."""
    poses, disps = poses.clone(), disps.clone()
    N, ht, wd = disps.shape
    P = ht * wd
    intr8 = intrinsics.reshape(-1)[:4] / 8.0
    wgt = weight * weight_scale
    srcs = torch.unique(ii)
    # fixed poses = SOURCE indices outside [t0,t1) (buffer.py:462-465); every other pose an edge touches is free
    outside = torch.ones(N, dtype=torch.bool)
    outside[t0:t1] = False
    fixed = torch.zeros(N, dtype=torch.bool).index_fill_(0, srcs, True) & outside
    touched = torch.zeros(N, dtype=torch.bool).index_fill_(0, torch.cat([ii, jj]), True)
    slot = torch.full((N,), -1, dtype=torch.long)
    ids = torch.nonzero(touched & ~fixed)[:, 0]
    slot[ids] = torch.arange(ids.numel())
    n = 6 * ids.numel()
    has_sens = disps_sens.reshape(N, -1).sum(1) > 0
    by_src = {int(k): torch.nonzero(ii == k)[:, 0] for k in srcs}
    for _ in range(n_iters):
        coords, valid, Ji, Jj, Jz = _linearise(poses, disps, intr8, ii, jj, ht, wd)
        r = coords - target
        w = valid[..., None] * wgt
        H = torch.zeros(n, n, dtype=poses.dtype)
        g = torch.zeros(n, dtype=poses.dtype)
        wJi, wJj = Ji * w[..., None], Jj * w[..., None]
        blocks = {"ii": torch.einsum("epcd,epcf->edf", wJi, Ji), "ij": torch.einsum("epcd,epcf->edf", wJi, Jj),
                  "jj": torch.einsum("epcd,epcf->edf", wJj, Jj)}
        vi, vj = -torch.einsum("epcd,epc->ed", wJi, r), -torch.einsum("epcd,epc->ed", wJj, r)
        Hv = H.view(n // 6, 6, n // 6, 6)
        gv = g.view(n // 6, 6)
        si, sj = slot[ii], slot[jj]
        for e in range(ii.shape[0]):
            a, b = int(si[e]), int(sj[e])
            if a >= 0:
                Hv[a, :, a] += blocks["ii"][e]
                gv[a] += vi[e]
            if b >= 0:
                Hv[b, :, b] += blocks["jj"][e]
                gv[b] += vj[e]
            if a >= 0 and b >= 0:
                Hv[a, :, b] += blocks["ij"][e]
                Hv[b, :, a] += blocks["ij"][e].T
        wz = w * Jz
        Ce = (wz * Jz).sum(-1)  # [E,P]
        we = -(wz * r).sum(-1)
        Ei = torch.einsum("epcd,epc->edp", Ji, wz)  # [E,6,P]
        Ej = torch.einsum("epcd,epc->edp", Jj, wz)
        d = H.diagonal()
        d += pose_ep + pose_damping * d  # solver.py:161-164, matrix.py:179-192
        S, gs = H.clone(), g.clone()
        Sv, gsv = S.view(n // 6, 6, n // 6, 6), gs.view(n // 6, 6)
        dz_parts = {}
        for k, es in by_src.items():
            C = Ce[es].sum(0)
            wk = we[es].sum(0)
            if bool(has_sens[k]):  # terms.py:258-268
                C = C + alpha
                wk = wk - alpha * (disps[k].reshape(-1) - disps_sens[k].reshape(-1))
            C = C + 1e-7 + (0.2 * eta[k].reshape(-1) + 1e-7)  # buffer.py:482-489
            Q = 1.0 / C
            rows, owners = [], []
            if int(slot[k]) >= 0:
                rows.append(Ei[es].sum(0))
                owners.append(int(slot[k]))
            for e in es.tolist():
                if int(sj[e]) >= 0:
                    rows.append(Ej[e])
                    owners.append(int(sj[e]))
            if rows:
                Ek = torch.cat(rows, 0)  # [6m, P]
                EQ = Ek * Q
                G = EQ @ Ek.T
                gq = EQ @ wk
                m = len(owners)
                Gv = G.view(m, 6, m, 6)
                for x in range(m):
                    gsv[owners[x]] -= gq[6 * x:6 * x + 6]
                    for y in range(m):
                        Sv[owners[x], :, owners[y]] -= Gv[x, :, y]
            dz_parts[k] = (Q, wk, rows, owners)
        dx = torch.linalg.solve(S, gs) if n else gs
        dxv = dx.view(-1, 6)
        for k, (Q, wk, rows, owners) in dz_parts.items():
            rhs = wk.clone()
            for x, own in enumerate(owners):
                rhs -= rows[x].T @ dxv[own]
            dz = rhs * Q
            dz = torch.where(dz > 10, torch.zeros_like(dz), dz)  # retractor.py:41
            disps[k] += dz.view(ht, wd)
        if n:
            poses[ids] = se3_mul(se3_exp(dxv), poses[ids])  # retractor.py:27-31
    disps.clamp_(min=1e-3)  # buffer.py:525
    return poses, disps
