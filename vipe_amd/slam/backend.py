"""Global bundle adjustment over all keyframes - host-side mirror of `SLAMBackend`
(vipe/slam/components/backend.py:31-122): a fresh non-incremental FactorGraph (max_factors 16 t), proximity edges
with the backend thresholds, then `steps` passes of `FactorGraph.update_batch` (hot loop B; 8 Gauss-Newton iterations
each, 16 when intrinsics / rig rotation are optimised).  Same constants as configs/slam/default.yaml.  The depth-prior
branch (`_iterate_with_depth`: half of the passes, then the sensor disparities are refreshed for the intrinsics found so
far, then the rest with the intrinsics held) is mirrored with a PLUGGABLE depth model (`GraphBuffer.update_disps_sens`);
the reference's monocular depth networks themselves are outside the path."""
from dataclasses import dataclass

import torch

from .factor_graph import FactorGraph


@dataclass
class BackendArgs:
    """configs/slam/default.yaml"""
    beta: float = 0.3
    backend_thresh: float = 22.0
    backend_radius: int = 2
    backend_nms: int = 3
    backend_iters: int = 24
    optimize_intrinsics: bool = False
    optimize_rig_rotation: bool = False
    cross_view: bool = True
    adaptive_cross_view: bool = False


class SLAMBackend:
    depth_model = None

    def __init__(self, update_module, video, args: BackendArgs, device):
        self.net, self.video, self.args, self.device = update_module, video, args, device
        self.last_graph = None

    @torch.no_grad()
    def run(self, steps=12, update_depth=True, log=False):
        """main update (fresh graph, GRU state re-read from the buffer) - backend.py:73-117"""
        a = self.args
        t = self.video.n_frames
        graph = FactorGraph(self.net, self.video, self.device, max_factors=16 * t, incremental=False,
                            cross_view=a.cross_view)
        graph.add_proximity_factors(rad=a.backend_radius, nms=a.backend_nms, thresh=a.backend_thresh, beta=a.beta)
        if a.adaptive_cross_view:
            self.video.build_adaptive_cross_view_idx()
        if len(graph.ii) > 0:
            more_iters = a.optimize_intrinsics or a.optimize_rig_rotation
            itrs = 16 if more_iters else 8
            if self.depth_model is not None:  # backend.py:45-63
                pre = steps // 2
                graph.update_batch(itrs=itrs, steps=pre, optimize_intrinsics=a.optimize_intrinsics,
                                   optimize_rig_rotation=a.optimize_rig_rotation)
                self.video.update_disps_sens(self.depth_model, frame_idx=None)
                graph.update_batch(itrs=itrs, steps=steps - pre, optimize_intrinsics=False,  # not the intrinsics again
                                   optimize_rig_rotation=a.optimize_rig_rotation)
            else:
                graph.update_batch(itrs=itrs, steps=steps, optimize_intrinsics=a.optimize_intrinsics,
                                   optimize_rig_rotation=a.optimize_rig_rotation)
        else:  # a single keyframe: take the sensor depth where there is one (backend.py:105-111)
            self.video.disps[0] = torch.where(self.video.disps_sens[0] > 0, self.video.disps_sens[0], self.video.disps[0])
        self.video.dirty[:t] = True
        self.last_graph = graph
        return graph

    @torch.no_grad()
    def run_if_necessary(self, steps=12, log=False):
        if self.args.optimize_intrinsics or self.args.optimize_rig_rotation:
            return self.run(steps=steps, update_depth=True, log=log)
