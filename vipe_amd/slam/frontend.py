"""Keyframe frontend: the scheduling around the update iteration (host-side mirror of
vipe/slam/components/frontend.py:32-159).  Called once per new keyframe: proximity edges, `iters1` update iterations,
keyframe-distance check (drop the second newest keyframe when it barely moved), `iters2` more iterations, pose
extrapolation for the next frame.  Same constants as the reference (max_factors 48, max_age 25, iters 4 + 2) and
configs/slam/default.yaml for the thresholds.  Everything numeric runs through FactorGraph / GraphBuffer (HIP)."""

from dataclasses import dataclass

import torch

from .._lib import upload, upload_many
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
        self._prox = None  # prefetched frame distances for the next keyframe's edge proposal, see _prefetch_proximity

    def _init_pose(self):
        """frontend.py:70-76: constant-velocity extrapolation, half the last relative motion."""
        assert self.t1 > 1
        p1, p2 = SE3(self.video.poses[self.t1 - 2][None]), SE3(self.video.poses[self.t1 - 1][None])
        w = (p2 * p1.inv()).log() * 0.5
        self.video.poses[self.t1] = (SE3.exp(w) * p2).data[0]

    def _next_frame(self, n_mean):
        """frontend.py:118-122 / 147-151 in ONE launch (`vipe_frontend_next_frame`): the constant-velocity pose of the frame
        that will be appended next (unless the caller supplies poses) and its disparity = the mean of the last `n_mean`
        keyframes' maps, per view.  CPU buffers (tests of the scheduling logic) take the torch formulation."""
        v = self.video
        if not v.poses.is_cuda:
            if not self.args.has_init_pose:
                self._init_pose()
            for q in range(v.n_views):
                v.disps[self.t1, q] = v.disps[self.t1 - n_mean:self.t1, q].mean()
            return
        from .._lib import check, lib, ptr, stream_ptr
        check(lib().vipe_frontend_next_frame(ptr(v.poses), ptr(v.disps), int(self.t1), int(v.n_views),
                                             int(v.disps.shape[2] * v.disps.shape[3]), int(n_mean),
                                             int(not self.args.has_init_pose), stream_ptr(v.poses)), "frontend_next_frame")

    def _iterate(self, n, **kw):
        for _ in range(n):
            self.graph.update(use_inactive=True, fixed_motion=self.args.has_init_pose, **kw)
            self.n_updates += 1

    def _prefetch_proximity(self):
        """The next keyframe's edge proposal (`add_proximity_factors` at the top of the next `_update`) needs the frame
        distances between the keyframes of the window INCLUDING the frame that will be appended - whose pose and
        disparity have just been initialised here (`_init_pose`, mean disparity).  Everything they depend on exists now,
        so the kernel is launched now and its result travels to pinned host memory behind the work already queued: the
        next `_update` finds it there instead of draining the stream for it (a ~1.5 ms bubble per keyframe).  Used only
        if nothing touched the geometry in between (`GraphBuffer.geom_version`) and exactly one frame was appended."""
        a = self.args
        if a.has_init_pose or not self.video.poses.is_cuda:  # the caller supplies the new frame's pose when it appends it:
            self._prox = None                                 # nothing to anticipate (host buffers: nothing to overlap)
            return
        t = self.t1 + 1
        t0, t1 = t - 5, max(t - a.frontend_window, 0)
        if t > self.video.poses.shape[0] or t0 < t1 or t0 >= t:
            self._prox = None
            return
        import numpy as np
        iin, jjn = np.meshgrid(np.arange(t0, t, dtype=np.int64), np.arange(t1, t, dtype=np.int64), indexing="ij")
        dev = self.video.device
        ii_d, jj_d = upload_many([iin.reshape(-1), jjn.reshape(-1)], dev)
        d = self.video.frame_distance_dense_disp(ii_d, jj_d, beta=a.beta, n_frames=t)
        d = d[:, 0] if d.shape[1] == 1 else d.mean(-1)
        host = torch.empty(d.shape, dtype=d.dtype, pin_memory=True)
        host.copy_(d, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._prox = dict(t=t, t0=t0, t1=t1, beta=a.beta, version=self.video.geom_version, host=host, event=ev)

    def _prefetched_distances(self, t0, t1, beta):
        p, self._prox = self._prox, None
        if (p is None or p["t"] != self.video.n_frames or p["t0"] != t0 or p["t1"] != t1 or p["beta"] != beta
                or p["version"] != self.video.geom_version):
            return None
        p["event"].synchronize()
        return p["host"].numpy()

    def _update(self):
        """frontend.py:78-129."""
        a = self.args
        self.t1 += 1
        if self.graph.corr is not None:
            self.graph.rm_factors(self.graph.host_edges()["age"] > self.max_age, store=True)  # host mirror: no read-back
        t0p, t1p = self.t1 - 5, max(self.t1 - a.frontend_window, 0)
        self.graph.add_proximity_factors(t0p, t1p, rad=a.frontend_radius, nms=a.frontend_nms, thresh=a.frontend_thresh,
                                         beta=a.beta, remove=True, dist=self._prefetched_distances(t0p, t1p, a.beta))
        self._iterate(self.iters1)
        dev = self.video.device
        ii_d, jj_d = upload_many([[self.t1 - 3], [self.t1 - 2]], dev)
        d = self.video.frame_distance_dense_disp(ii_d, jj_d, beta=a.beta,
                                                 bidirectional=True)
        if float(d.cpu().max()) < a.keyframe_thresh:  # one small copy (the read-back is the sync either way)
            self.graph.rm_second_newest_keyframe(self.t1 - 2)
            self.t1 -= 1
        else:
            self._iterate(self.iters2)
        self._next_frame(1)
        self.video.dirty[int(self.graph.host_edges()["ii"].min()):self.t1] = True  # frontend.py:124 (host mirror: no read-back)
        self.video.touch()
        self._prefetch_proximity()

    def _initialize(self):
        """frontend.py:131-155."""
        a = self.args
        self.t1 = self.video.n_frames
        self.graph.add_neighborhood_factors(0, self.t1, r=1 if a.seq_init else 3)
        self._iterate(8, t0=1)
        if not a.seq_init:
            self.graph.add_proximity_factors(0, 0, rad=2, nms=2, thresh=a.frontend_thresh, remove=False)
            self._iterate(8, t0=1)
        self._next_frame(4)
        self.video.dirty[:self.t1] = True
        self.is_initialized = True
        self.graph.rm_factors(self.graph.host_edges()["ii"] < a.warmup - 4, store=True)
        self.video.touch()
        self._prefetch_proximity()

    def run(self):
        """frontend.py:157-167: call after every keyframe appended to the buffer."""
        if not self.is_initialized and self.video.n_frames == self.args.warmup:
            self._initialize()
        elif self.is_initialized and self.t1 < self.video.n_frames:
            self._update()
