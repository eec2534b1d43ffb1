"""Sparse keypoint tracks between a tracker and the dense path - mirror of the `SparseTracks` base class
(vipe/slam/components/sparse_tracks/__init__.py:27-141).  The trackers themselves (cuVSLAM, SuperPoint) are outside the
path: a subclass fills `observations` in `track_image`; `ReplayedSparseTracks` takes them from the caller.

What the dense path reads: `get_correspondences` / `get_observations` (the motion filter's track score) and
`compute_dense_disp_target_weight` (the second flow term of the BA, buffer.py:422-447): per edge the flow of the keypoints
seen in both frames, splatted bilinearly onto the 1/8 grid.  The reference loops over the edges on the host with eight
`index_add_` launches and two device uploads per edge; here the correspondences of ALL edges are gathered on the host
once, uploaded once, and splatted by two launches of the library's atomic scatter kernel."""
import numpy as np
import torch

from ..ext import scatter_ext


class SparseTracks:
    """Single-camera tracks, as the reference: observations[view][frame] = {keypoint id: (u, v) in image pixels}."""
    enabled = True

    def __init__(self, n_views):
        self.observations = [[] for _ in range(n_views)]

    def track_image(self, frame_data_list):
        raise NotImplementedError("a tracker fills `observations`; see ReplayedSparseTracks")

    def get_correspondences(self, view_idx, source_frame_idx, target_frame_idx):
        """ids of the keypoints observed in both frames (ascending)"""
        a, b = self.observations[view_idx][int(source_frame_idx)], self.observations[view_idx][int(target_frame_idx)]
        return torch.tensor(sorted(a.keys() & b.keys()), dtype=torch.long)

    def get_observations(self, view_idx, frame_idx, keypoint_indices):
        if len(keypoint_indices) == 0:
            return torch.empty(0, 2, device=keypoint_indices.device)
        uvs = self.observations[view_idx][int(frame_idx)]
        return torch.tensor(np.stack([uvs[int(k)] for k in keypoint_indices.cpu().numpy()], 0)).to(keypoint_indices.device).float()

    def compute_dense_disp_target_weight(self, source_view_inds, source_frame_inds, target_view_inds, target_frame_inds,
                                         image_size, dense_disp_size):
        """-> target [E,h,w,2] (grid position + mean track flow of the cell), weight [E,h,w,2] (splatted bilinear weights,
        cells below 0.1 cleared).  sparse_tracks/__init__.py:68-141 with `bilinear_splatting_inplace` (utils/depth.py:123-155):
        corner (floor(u + 0.5), floor(v + 0.5)) and its right / lower neighbours, weights from the offset to that corner -
        which lies in [-0.5, 0.5), so they are not the usual bilinear weights - points whose four corners are not all
        inside the grid are dropped."""
        device = source_view_inds.device
        E, (h, w) = len(source_view_inds), dense_disp_size
        assert E == len(target_view_inds) == len(source_frame_inds) == len(target_frame_inds)
        sv, sf = source_view_inds.cpu().numpy(), source_frame_inds.cpu().numpy()
        tv, tf = target_view_inds.cpu().numpy(), target_frame_inds.cpu().numpy()
        terms, src, flow = [], [], []
        for e in range(E):
            assert sv[e] == tv[e], "Only same view tracking is supported"
            a, b = self.observations[sv[e]][int(sf[e])], self.observations[tv[e]][int(tf[e])]
            ids = sorted(a.keys() & b.keys())
            if ids:
                s = np.asarray([a[k] for k in ids], np.float32).reshape(-1, 2)
                terms.append(np.full(len(ids), e, np.int64))
                src.append(s)
                flow.append(np.asarray([b[k] for k in ids], np.float32).reshape(-1, 2) - s)
        value = torch.zeros(E * h * w, 2, device=device)
        weight = torch.zeros(E * h * w, device=device)
        if terms:
            fac = torch.tensor([w / image_size[1], h / image_size[0]], device=device)
            term = torch.from_numpy(np.concatenate(terms)).to(device)
            uv = torch.from_numpy(np.concatenate(src)).to(device) * fac
            data = torch.from_numpy(np.concatenate(flow)).to(device) * fac
            u, v = uv.unbind(-1)
            x0, y0 = torch.floor(u + 0.5).long(), torch.floor(v + 0.5).long()
            ok = (x0 >= 0) & (x0 + 1 < w) & (y0 >= 0) & (y0 + 1 < h)
            term, data, u, v, x0, y0 = term[ok], data[ok], u[ok], v[ok], x0[ok], y0[ok]
            wx, wy = u - x0.float(), v - y0.float()
            base = term * (h * w) + y0 * w + x0
            idx = torch.cat([base, base + w, base + 1, base + w + 1])                       # (x0,y0) (x0,y1) (x1,y0) (x1,y1)
            cw = torch.cat([(1 - wx) * (1 - wy), (1 - wx) * wy, wx * (1 - wy), wx * wy])
            if idx.numel():
                scatter_ext.scatter_sum(data.repeat(4, 1) * cw[:, None], idx, 0, value, None)
                scatter_ext.scatter_sum(cw.contiguous(), idx, 0, weight, None)
        value = (value / weight[:, None]).view(E, h, w, 2)
        weight = weight.view(E, h, w, 1).repeat(1, 1, 1, 2)
        value[torch.isnan(value)] = 0.0
        value[weight < 0.1] = 0.0
        weight[weight < 0.1] = 0.0
        yy, xx = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
        value[..., 0] += xx
        value[..., 1] += yy
        return value, weight


class DummySparseTracks(SparseTracks):
    """sparse_tracks/__init__.py:144-149: no tracker"""
    enabled = False

    def track_image(self, frame_data_list):
        for obs in self.observations:
            obs.append({})


class ReplayedSparseTracks(SparseTracks):
    """Tracks computed elsewhere: `tracks[view][frame]` = {keypoint id: (u, v)}, handed over frame by frame."""

    def __init__(self, tracks):
        super().__init__(len(tracks))
        self._tracks = tracks

    def track_image(self, frame_data_list):
        for v, obs in enumerate(self.observations):
            obs.append(dict(self._tracks[v][len(obs)]))
