"""Poses (and optionally dense disparities) of the NON-keyframe frames - host-side mirror of `InnerFiller`
(vipe/slam/components/inner_filler.py:46-138), the second pass of `SLAMSystem.run` (system.py:284-294): every frame of the
video is appended behind the keyframes, in chunks of `infill_chunk_size`; a chunk's poses start from the constant-velocity
interpolation between its neighbouring keyframes and are refined by ten update iterations on a graph of (keyframe ->
frame) edges - motion only (`motion_only=True, limited_disp=True`) unless `infill_dense_disp`.  The buffer's frame
count returns to the keyframes after every chunk."""
from dataclasses import dataclass

import torch

from ..ext import lietorch as lt
from ..ext.lietorch import SE3
from .factor_graph import FactorGraph


@dataclass
class InfillArgs:
    """configs/slam/default.yaml:36-39"""
    infill_chunk_size: int = 16
    infill_dense_disp: bool = False


@dataclass
class FilledReturn:
    poses: SE3                       # world -> camera of every frame (inverse of c2w)
    dense_disps: torch.Tensor = None

    def scale(self, factor):
        self.poses.data[..., :3] *= factor
        if self.dense_disps is not None:
            self.dense_disps /= factor


class InnerFiller:
    def __init__(self, update_module, video, args: InfillArgs, device):
        self.net, self.video, self.args, self.device = update_module, video, args, device
        self.start_idx = -1
        self.filled_poses, self.filled_dense_disps = [], []

    def set_start_idx(self, start_idx):
        self.start_idx = int(start_idx)

    def check(self):
        assert self.start_idx >= 0
        return self.video.n_frames - self.start_idx >= self.args.infill_chunk_size

    @torch.no_grad()
    def interpolate(self):
        """inner_filler.py:62-76: (t0, t1, poses) - for every appended frame its left (inclusive) nearest keyframe, that
        keyframe's successor, and the constant-velocity pose Exp(log(G_t1 G_t0^-1) dt / DT) G_t0."""
        v, s = self.video, self.start_idx
        total = v.n_frames
        m_tstamp, n_tstamp = v.tstamp[s:total], v.tstamp[:s]
        t0 = torch.searchsorted(n_tstamp, m_tstamp, right=True) - 1
        t1 = torch.where(t0 < s - 1, t0 + 1, t0)
        d_time = n_tstamp[t1] - n_tstamp[t0] + 1e-3  # frames beyond the last keyframe: zero velocity
        n_pose = SE3(v.poses[:s])
        vel = (n_pose[t1] * n_pose[t0].inv()).log() / d_time.unsqueeze(-1)
        w = vel * (m_tstamp - n_tstamp[t0]).unsqueeze(-1)
        return t0, t1, SE3.exp(w) * n_pose[t0]

    @torch.no_grad()
    def compute(self):
        v, s = self.video, self.start_idx
        total = v.n_frames
        t0, t1, m_pose = self.interpolate()
        v.poses[s:total] = m_pose.data
        if self.args.infill_dense_disp:
            v.disps[s:total] = v.disps[t0].mean(dim=[2, 3], keepdim=True)
            v.disps[s:total] = torch.where(v.disps_sens[s:total] > 0, v.disps_sens[s:total], v.disps[s:total])
        v.touch()
        graph = FactorGraph(self.net, v, self.device, max_factors=-1, incremental=True, cross_view=False)
        infill_inds = torch.arange(s, total, device=self.device)
        graph.add_factors(t0, infill_inds)
        graph.add_factors(t1, infill_inds)
        if self.args.infill_dense_disp:
            graph.add_factors(infill_inds, t0)
            graph.add_factors(infill_inds, t1)
        for _ in range(10):
            graph.update(s, total, motion_only=not self.args.infill_dense_disp, limited_disp=True)
        self.filled_poses.append(SE3(v.poses[s:total].clone()))
        if self.args.infill_dense_disp:
            self.filled_dense_disps.append(v.disps[s:total].clone())
        v.n_frames = s
        self.last_graph = graph

    def get_result(self):
        return FilledReturn(poses=lt.cat(self.filled_poses, dim=0),
                            dense_disps=torch.cat(self.filled_dense_disps, dim=0) if self.filled_dense_disps else None)
