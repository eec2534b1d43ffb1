"""CPU restatement (test infrastructure) of the sparse-track target / weight maps the BA's second flow term reads:
`SparseTracks.compute_dense_disp_target_weight` (vipe/slam/components/sparse_tracks/__init__.py:68-141) over
`bilinear_splatting_inplace` (vipe/utils/depth.py:123-155).  Plain numpy, one edge and one corner at a time.
The splat is pinned by outputs of the reference function itself (tests/golden/splat_reference.npz)."""
import numpy as np


def bilinear_splat(data, uv, out_data, out_weight):
    """depth.py:123-155: corner = floor(uv + 0.5); weights from the offset to that corner (in [-0.5, 0.5)); points whose
    2 x 2 corners are not all inside are dropped.  In place on out_data [H,W,V] / out_weight [H,W]."""
    H, W, _ = out_data.shape
    u, v = uv[:, 0], uv[:, 1]
    x0, y0 = np.floor(u + 0.5).astype(np.int64), np.floor(v + 0.5).astype(np.int64)
    ok = (x0 >= 0) & (x0 + 1 < W) & (y0 >= 0) & (y0 + 1 < H)
    x0, y0, data, u, v = x0[ok], y0[ok], data[ok], u[ok], v[ok]
    wx, wy = u - x0.astype(np.float32), v - y0.astype(np.float32)
    for dx, dy, cw in ((0, 0, (1 - wx) * (1 - wy)), (0, 1, (1 - wx) * wy), (1, 0, wx * (1 - wy)), (1, 1, wx * wy)):
        np.add.at(out_data, (y0 + dy, x0 + dx), data * cw[:, None])
        np.add.at(out_weight, (y0 + dy, x0 + dx), cw)


def dense_disp_target_weight(observations, view_inds, source_frames, target_frames, image_size, dense_disp_size):
    """sparse_tracks/__init__.py:68-141 -> target [E,h,w,2], weight [E,h,w,2]"""
    h, w = dense_disp_size
    fac = np.array([w / image_size[1], h / image_size[0]], np.float32)
    E = len(view_inds)
    value, weight = np.zeros((E, h, w, 2), np.float32), np.zeros((E, h, w), np.float32)
    for e in range(E):
        a, b = observations[view_inds[e]][source_frames[e]], observations[view_inds[e]][target_frames[e]]
        ids = sorted(set(a) & set(b))
        if not ids:
            continue
        s = np.asarray([a[k] for k in ids], np.float32)
        t = np.asarray([b[k] for k in ids], np.float32)
        bilinear_splat((t - s) * fac, s * fac, value[e], weight[e])
    with np.errstate(divide="ignore", invalid="ignore"):
        value = value / weight[..., None]
    weight = np.repeat(weight[..., None], 2, -1)
    value[np.isnan(value)] = 0.0
    value[weight < 0.1] = 0.0
    weight[weight < 0.1] = 0.0
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    value[..., 0] += xx
    value[..., 1] += yy
    return value, weight
