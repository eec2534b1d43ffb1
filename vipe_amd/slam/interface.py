"""SLAM map container - host-side mirror of `SLAMMap` (vipe/slam/interface.py:25-141): the filtered dense-disparity
point cloud that `GraphBuffer.extract_slam_map` (buffer.py:595-645) produces after the update path, and its projection
into a target camera for depth alignment.  SURVEY 8(f) row 3."""
from dataclasses import dataclass

import numpy as np
import torch

from .._lib import require


@dataclass(kw_only=True)
class SLAMMap:
    dense_disp_xyz: torch.Tensor       # (M, 3)
    dense_disp_rgb: torch.Tensor       # (M, 3) RGB 0-1
    dense_disp_packinfo: torch.Tensor  # (N, V, 2) [start, count] per keyframe and view
    dense_disp_frame_inds: list        # frame index of each keyframe, sorted

    def scale(self, factor):
        self.dense_disp_xyz *= factor

    @staticmethod
    def from_masked_dense_disp(xyz, rgb, mask, tstamps):
        """xyz, rgb (N,V,H,W,3); mask (N,V,H,W); tstamps (N,)  (interface.py:40-60)"""
        assert torch.all(tstamps[1:] > tstamps[:-1]), "Timestamps should be sorted."
        N, V, H, W, C = xyz.shape
        flat = mask.reshape(-1)
        valid_count = mask.sum([2, 3]).reshape(-1)
        packinfo = torch.stack([torch.cumsum(valid_count, 0) - valid_count, valid_count], dim=-1).reshape(N, V, 2)
        return SLAMMap(dense_disp_xyz=xyz.reshape(-1, C)[flat], dense_disp_rgb=rgb.reshape(-1, C)[flat],
                       dense_disp_packinfo=packinfo, dense_disp_frame_inds=tstamps.tolist())

    def get_dense_disp_pcd(self, keyframe_idx, view_idx=-1):
        if view_idx == -1:
            parts = [self.get_dense_disp_pcd(keyframe_idx, v) for v in range(self.dense_disp_packinfo.shape[1])]
            return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)
        start, count = [int(x) for x in self.dense_disp_packinfo[keyframe_idx, view_idx]]
        return self.dense_disp_xyz[start:start + count], self.dense_disp_rgb[start:start + count]

    def get_dense_disp_full_pcd(self):
        parts = [self.get_dense_disp_pcd(k) for k in range(len(self.dense_disp_frame_inds))]
        return torch.cat([p[0] for p in parts], 0), torch.cat([p[1] for p in parts], 0)

    def project_map(self, frame_tstamp, view_idx, target_size, target_intrinsics, target_pose, infill=False, tstamp_nn=3):
        """Depth image [H,W] of the points of the keyframes around `frame_tstamp` seen from the camera whose
        camera->world pose is `target_pose` (an SE3), through a pinhole camera (interface.py:92-141).  Where several points fall into one pixel the
        reference keeps whichever its scatter writes last (unspecified); here the nearest one is kept.  `infill`: every
        pixel takes the depth of the projected point nearest to its centre (`utils_ext.nearest_neighbours`,
        interface.py:126-139)."""
        right = int(np.searchsorted(self.dense_disp_frame_inds, frame_tstamp))
        right = min(right + tstamp_nn, len(self.dense_disp_frame_inds) - 1)
        left = max(right - 2 * tstamp_nn, 0)
        xyz = torch.cat([self.get_dense_disp_pcd(k, view_idx)[0] for k in range(left, right + 1)], 0)
        # the reference transforms with target_pose.inv().matrix(): target_pose is camera->world there
        T = target_pose.inv().matrix()
        xyz = xyz @ T[:3, :3].T + T[:3, 3]
        fx, fy, cx, cy = [float(v) for v in target_intrinsics[:4]]
        z = xyz[:, 2]  # limit_min_depth=False: no clamp before the division (cameras.py:175-177)
        uu, vv = fx * xyz[:, 0] / z + cx, fy * xyz[:, 1] / z + cy
        H, W = target_size
        ok = (uu > 0) & (uu < W) & (vv > 0) & (vv < H) & (1.0 / z > 0)
        uu, vv, depth = uu[ok], vv[ok], z[ok]
        if infill:
            from ..ext import utils_ext
            tree = torch.stack((uu, vv), dim=-1).contiguous()
            qx, qy = torch.meshgrid(torch.arange(W, device=xyz.device).float() + 0.5,
                                    torch.arange(H, device=xyz.device).float() + 0.5, indexing="xy")
            query = torch.stack((qx, qy), dim=-1).reshape(-1, 2).contiguous()
            _, inds = utils_ext.nearest_neighbours(query, tree, 1)
            return depth[inds.view(-1).long()].reshape(H, W)
        out = torch.full((H * W,), float("inf"), device=xyz.device)
        out.scatter_reduce_(0, vv.floor().long() * W + uu.floor().long(), depth, reduce="amin")
        return torch.where(torch.isinf(out), torch.zeros_like(out), out).view(H, W)


class SLAMOutput:
    """interface.py:143-163: what `SLAMSystem.run` returns - trajectory (camera -> world per frame, SE3 [N]), intrinsics
    [V,4], the rig, the map, the BA residual."""

    def __init__(self, *, trajectory, intrinsics, rig=None, slam_map=None, ba_residual=0.0):
        self.trajectory, self.intrinsics, self.rig, self.slam_map, self.ba_residual = trajectory, intrinsics, rig, slam_map, ba_residual

    @property
    def keyframe_ids(self):
        assert self.slam_map is not None, "SLAM map not available."
        return np.array(self.slam_map.dense_disp_frame_inds)

    def get_view_trajectory(self, view_idx):
        assert self.rig is not None, "Rig not available."
        return self.trajectory * self.rig[view_idx][None]

