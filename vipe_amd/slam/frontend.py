"""Keyframe frontend: the scheduling around the update iteration (host-side mirror of
vipe/slam/components/frontend.py:32-159).  Called once per new keyframe: proximity edges, `iters1` update iterations,
keyframe-distance check (drop the second newest keyframe when it barely moved), `iters2` more iterations, pose
extrapolation for the next frame.  Same constants as the reference (max_factors 48, max_age 25, iters 4 + 2) and
configs/slam/default.yaml for the thresholds.  Everything numeric runs through FactorGraph / GraphBuffer (HIP)."""

from dataclasses import dataclass

import torch

from ..ext.lietorch import SE3
from .factor_graph import FactorGraph


@dataclass
class FrontendArgs:
    """configs/slam/default.yaml"""
    warmup: int = 8
    beta: float = 0.3
    keyframe_thresh: float = 4.0
    frontend_thresh: float = 16.0
    frontend_window: int = 25
    frontend_radius: int = 2
    frontend_nms: int = 1
    seq_init: bool = True
    has_init_pose: bool = False
    cross_view: bool = True


class SLAMFrontend:
    def __init__(self, update_module, video, args: FrontendArgs, device):
        self.video = video
        self.graph = FactorGraph(update_module, video, device, max_factors=48, incremental=True, cross_view=args.cross_view)
        self.t1 = 0
        self.is_initialized = False
        self.max_age, self.iters1, self.iters2 = 25, 4, 2
        self.args = args
        self.n_updates = 0

    def _init_pose(self):
        """frontend.py:70-76: constant-velocity extrapolation, half the last relative motion."""
        assert self.t1 > 1
        p1, p2 = SE3(self.video.poses[self.t1 - 2][None]), SE3(self.video.poses[self.t1 - 1][None])
        w = (p2 * p1.inv()).log() * 0.5
        self.video.poses[self.t1] = (SE3.exp(w) * p2).data[0]

    def _iterate(self, n, **kw):
        for _ in range(n):
            self.graph.update(use_inactive=True, fixed_motion=self.args.has_init_pose, **kw)
            self.n_updates += 1

    def _update(self):
        """frontend.py:78-129."""
        a = self.args
        self.t1 += 1
        if self.graph.corr is not None:
            self.graph.rm_factors(self.graph.host_edges()["age"] > self.max_age, store=True)  # host mirror: no read-back
        self.graph.add_proximity_factors(self.t1 - 5, max(self.t1 - a.frontend_window, 0), rad=a.frontend_radius,
                                         nms=a.frontend_nms, thresh=a.frontend_thresh, beta=a.beta, remove=True)
        self._iterate(self.iters1)
        dev = self.video.device
        d = self.video.frame_distance_dense_disp(torch.tensor([self.t1 - 3], device=dev),
                                                 torch.tensor([self.t1 - 2], device=dev), beta=a.beta, bidirectional=True)
        if d.max().item() < a.keyframe_thresh:
            self.graph.rm_second_newest_keyframe(self.t1 - 2)
            self.t1 -= 1
        else:
            self._iterate(self.iters2)
        if not a.has_init_pose:
            self._init_pose()
        for v in range(self.video.n_views):
            self.video.disps[self.t1, v] = self.video.disps[self.t1 - 1, v].mean()

    def _initialize(self):
        """frontend.py:131-155."""
        a = self.args
        self.t1 = self.video.n_frames
        self.graph.add_neighborhood_factors(0, self.t1, r=1 if a.seq_init else 3)
        self._iterate(8, t0=1)
        if not a.seq_init:
            self.graph.add_proximity_factors(0, 0, rad=2, nms=2, thresh=a.frontend_thresh, remove=False)
            self._iterate(8, t0=1)
        if not a.has_init_pose:
            self._init_pose()
        for v in range(self.video.n_views):
            self.video.disps[self.t1, v] = self.video.disps[self.t1 - 4:self.t1, v].mean()
        self.is_initialized = True
        self.graph.rm_factors(self.graph.host_edges()["ii"] < a.warmup - 4, store=True)

    def run(self):
        """frontend.py:157-167: call after every keyframe appended to the buffer."""
        if not self.is_initialized and self.video.n_frames == self.args.warmup:
            self._initialize()
        elif self.is_initialized and self.t1 < self.video.n_frames:
            self._update()
