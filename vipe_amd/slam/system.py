"""`SLAMSystem` - host-side mirror of the reference's top-level driver (vipe/slam/system.py:51-316) over frames that are
already decoded, resized and resident on the device (decoding / resizing / the monocular depth networks / sparse tracks /
the rerun visualisation are outside the path):

    pass 1  every frame through the motion filter; accepted frames (and the last one) become keyframes - features and
            context into the buffer, sensor disparities from the frame's metric depth or a caller-supplied depth model,
            poses from the frame when given - then `SLAMFrontend.run()`; `SLAMBackend.run_if_necessary(5)` at
            `frontend_backend_iters` keyframes (system.py:236-273).  Scheduled as a two-stage pipeline
            (`SLAMConfig.pipeline_filter`): the filter of frame f+1 - feature encoder + one application of the update
            operator against the last keyframe, which depend on nothing the frontend changes - is enqueued on a side stream
            BEFORE keyframe f is optimised on the main stream and its score is collected afterwards
            (`MotionFilter.prefetch` / `check`); same decisions, same results, the filter fills the chip the frontend's
            single-workgroup solves leave idle
            `SLAMBackend.run(7)`, `SLAMBackend.run(backend_iters)` (system.py:279-282)
    pass 2  every frame appended again behind the keyframes, `InnerFiller` in chunks (system.py:284-294)
            `extract_slam_map`, `SLAMOutput(trajectory = filled poses inverted, intrinsics, rig, map)` (system.py:303-316)
"""
from dataclasses import dataclass, field

import torch

from ..ext.lietorch import SE3
from .backend import BackendArgs, SLAMBackend
from .buffer import GraphBuffer
from .frontend import FrontendArgs, SLAMFrontend
from .inner_filler import InfillArgs, InnerFiller
from .interface import SLAMOutput
from .motion_filter import DroidNet, MotionFilter


@dataclass
class SLAMConfig:
    """configs/slam/default.yaml (the fields the path reads)"""
    buffer: int = 1024
    filter_thresh: float = 2.4
    init_disp: float = 1.0
    map_filter_thresh: float = 0.05
    frontend_backend_iters: tuple = (16, 64, 256)
    cross_view_idx: object = None
    frontend: FrontendArgs = field(default_factory=FrontendArgs)
    backend: BackendArgs = field(default_factory=BackendArgs)
    infill: InfillArgs = field(default_factory=InfillArgs)
    # not in the reference's configuration - how THIS build schedules pass 1 (results do not depend on either):
    pipeline_filter: bool = True   # motion filter of frame f+1 on a side stream under keyframe f's frontend step
    pause_gc: bool = True          # keep the cyclic collector off during run() (its gen-2 sweeps cost up to 0.4 s per clip)
    release_cached_memory: bool = False  # several clips share this card (one process each): hand the global BA's pyramid
                                         # blocks (~100 GB at 200 keyframes) back to the driver after the second backend pass
                                         # instead of keeping them cached for the next clip of this process
    backend_lock_path: str = None        # ... and let their global-BA phases - each fills the chip by itself and wants the
                                         # pyramid budget to itself - take turns (an advisory file lock held around the two
                                         # backend passes), while the other clips' pass 1 / pass 2 run beside it


@dataclass
class Frame:
    """What the path reads of the reference's `VideoFrame` (one per view): rgb [H,W,3] float 0-1, optional mask [H,W]
    bool (True = usable pixel, system.py:199-211), metric depth [H,W], intrinsics [4] (or [5] MEI) at frame resolution,
    camera->world pose (SE3)."""
    rgb: torch.Tensor
    mask: torch.Tensor = None
    metric_depth: torch.Tensor = None
    intrinsics: torch.Tensor = None
    pose: SE3 = None


class SLAMSystem:
    def __init__(self, device=torch.device("cuda"), config=None, droid_net=None, depth_model=None, sparse_tracks=None,
                 motion_filter_cls=MotionFilter):
        self.device, self.config = device, config or SLAMConfig()
        self.motion_filter_cls = motion_filter_cls  # pluggable keyframe selection (same constructor / check / prefetch)
        self.droid_net = droid_net if droid_net is not None else DroidNet()
        self.metric_depth = depth_model  # system.py:114-127 builds it from `keyframe_depth`; here the caller's
        self.sparse_tracks = sparse_tracks
        # system.py:91 builds a tracker from the config; here the caller's (`track_image(frames)` per frame of pass 1,
        # `enabled`, `compute_dense_disp_target_weight` for the BA's track term) or None = the reference's dummy

    def _build_components(self, height, width, n_views, rig, camera_type):
        c = self.config
        self.buffer = GraphBuffer(height, width, n_views=n_views, buffer_size=c.buffer, init_disp=c.init_disp,
                                  cross_view_idx=c.cross_view_idx, camera_type=camera_type, device=self.device)
        self.buffer.rig[:] = rig.data.to(self.device)
        self.buffer.sparse_tracks = self.sparse_tracks
        self.motion_filter = self.motion_filter_cls(self.droid_net, sparse_tracks=self.sparse_tracks, thresh=c.filter_thresh,
                                                    device=self.device)
        self.frontend = SLAMFrontend(self.droid_net.update, self.buffer, c.frontend, self.device)
        self.backend = SLAMBackend(self.droid_net.update, self.buffer, c.backend, self.device)
        self.inner_filler = InnerFiller(self.droid_net.update, self.buffer, c.infill, self.device)
        if self.metric_depth is not None:
            assert n_views == 1, "the global scale lies in the null space of a multi-view problem (system.py:115-118)"
        self.backend.depth_model = self.metric_depth

    def _precompute_features(self, frames):
        """system.py:194-218: images [V,3,H,W]; masks [V,h,w] True = INVALID (bilinear 1/8 downsample > 0.9, inverted)."""
        images = torch.stack([f.rgb for f in frames]).permute(0, 3, 1, 2).contiguous().to(self.device).float()
        masks = None
        if all(f.mask is not None for f in frames):
            ms = []
            for f in frames:
                mh, mw = f.mask.shape[0] // 8, f.mask.shape[1] // 8
                m = torch.nn.functional.interpolate(f.mask[None, None].float().to(self.device), (mh, mw), mode="bilinear")[0, 0] > 0.9
                ms.append(~m)
            masks = torch.stack(ms)
        return images, masks

    def _add_keyframe(self, frame_idx, images, buffer_masks, frames, phase, reuse=None):
        """system.py:131-165.  `reuse`: (fmap, net, inp) the motion filter has just computed for these very images."""
        b = self.buffer
        k = b.n_frames
        b.tstamp[k] = frame_idx
        b.images[k] = images
        if reuse is not None:
            b.fmaps[k], b.nets[k], b.inps[k] = reuse
        else:
            b.fmaps[k] = self.droid_net.encode_features(images)
            b.nets[k], b.inps[k] = self.droid_net.encode_context(images)
        if buffer_masks is not None:
            b.masks[k] = buffer_masks
        for v, f in enumerate(frames):
            if k == 0:
                assert f.intrinsics is not None, "the first frame must carry intrinsics"
                b.intrinsics[v] = f.intrinsics.to(self.device)
            if f.metric_depth is not None:
                d = f.metric_depth[3::8, 3::8].to(self.device)
                assert not self.config.backend.optimize_intrinsics
                b.disps_sens[k, v] = torch.where(d > 0, d.reciprocal(), d)
            if f.pose is not None and phase == 1:
                b.poses[k] = (SE3(b.rig[v]) * f.pose.inv()).data
                b.touch()  # geometry of slot k rewritten from outside: prefetched frame distances are stale
        if phase == 1:
            b.update_disps_sens(self.metric_depth, frame_idx=k)
        b.n_frames += 1

    def _run_passes(self, frames, total):
        import time
        b, mf = self.buffer, self.motion_filter
        on_gpu = torch.device(self.device).type == "cuda"
        main = torch.cuda.current_stream(self.device) if on_gpu else None
        side = None
        if on_gpu and self.config.pipeline_filter:
            side = getattr(self, "_filter_stream", None)
            if side is None or side.device != main.device:
                side = self._filter_stream = torch.cuda.Stream(device=self.device)

        def mark(name):  # phase boundaries: three stream drains per clip
            if on_gpu:
                torch.cuda.synchronize(self.device)
            self.timings[name] = time.perf_counter() - t_start

        from . import factor_graph as _fg
        work0 = dict(_fg.WORK)
        self.timings, self.work = {}, {}
        t_start = time.perf_counter()
        nxt = self._precompute_features(frames[0])
        for frame_idx, fl in enumerate(frames):  # SLAM pass 1/2 (system.py:236-273)
            images, masks = nxt
            if self.sparse_tracks is not None:
                self.sparse_tracks.track_image(fl)
            kept = mf.check(images, masks)  # collects the prefetched first half when there is one
            is_keyframe = kept or frame_idx == total - 1
            if is_keyframe:  # a frame the filter kept has its features and context there already
                self._add_keyframe(frame_idx, images, masks, fl, phase=1,
                                   reuse=(mf.f_fmap, mf.f_net, mf.f_inp) if kept else None)
            if frame_idx + 1 < total:
                nxt = self._precompute_features(frames[frame_idx + 1])
                if side is not None:  # frame f+1's filter runs beside keyframe f's optimisation
                    mf.prefetch(nxt[0], nxt[1], stream=side)
            self.frontend.run()
            # the backend in between corrects intrinsics / extrinsics early (system.py:269-272)
            if is_keyframe and b.n_frames in self.config.frontend_backend_iters:
                self.backend.run_if_necessary(5)
        if side is not None:
            main.wait_stream(side)
        mark("pass1_seconds")
        self.work["pass1"] = {k: _fg.WORK[k] - work0[k] for k in work0}
        self.n_keyframes = int(b.n_frames)
        release = on_gpu and self.config.release_cached_memory
        lock = None
        if self.config.backend_lock_path:
            import fcntl
            lock = open(self.config.backend_lock_path, "a")
            t_wait = time.perf_counter()
            fcntl.flock(lock, fcntl.LOCK_EX)
            self.timings["backend_lock_wait_seconds"] = time.perf_counter() - t_wait
        try:
            self.backend.run(7)
            gb = self.backend.run(self.config.backend.backend_iters, update_depth=False)
            self.backend_edges = int(gb.host_edges()["ii"].shape[0])
            if release:
                del gb
                self.backend.last_graph = None
                torch.cuda.synchronize(self.device)
                torch.cuda.empty_cache()
        finally:
            if lock is not None:
                if on_gpu:
                    torch.cuda.synchronize(self.device)  # the next holder gets the chip and the memory, not a queue behind ours
                fcntl.flock(lock, fcntl.LOCK_UN)
                lock.close()
        mark("global_ba_done_seconds")
        self.inner_filler.set_start_idx(b.n_frames)
        # SLAM pass 2/2 (system.py:284-294): every frame appended behind the keyframes, `InnerFiller.compute` whenever
        # `infill_chunk_size` of them are there (and at the last frame).  The frames of a chunk go through the two encoders
        # in ONE batched call (per image the same arithmetic; instance norm is per image): 39 launches per chunk instead of
        # 39 per frame, and at 16 images the encoder kernels fill the chip
        chunk = max(1, int(self.config.infill.infill_chunk_size))
        for c0 in range(0, total, chunk):
            idx = range(c0, min(c0 + chunk, total))
            pre = [self._precompute_features(frames[i]) for i in idx]
            feats = None
            if on_gpu and len(idx) > 1:
                from .encoders import normalize_images
                imgs = torch.cat([p[0] for p in pre], 0)
                x4 = normalize_images(imgs)
                fmap = self.droid_net.encode_features(imgs, x4)
                net, inp = self.droid_net.encode_context(imgs, x4)
                V = pre[0][0].shape[0]
                feats = [(fmap[j * V:(j + 1) * V], net[j * V:(j + 1) * V], inp[j * V:(j + 1) * V]) for j in range(len(idx))]
            for j, i in enumerate(idx):
                self._add_keyframe(i, pre[j][0], pre[j][1], frames[i], phase=2, reuse=feats[j] if feats else None)
                if self.inner_filler.check() or i == total - 1:
                    self.inner_filler.compute()
        mark("pass2_done_seconds")
        self.work["total"] = {k: _fg.WORK[k] - work0[k] for k in work0}

    @torch.no_grad()
    def run(self, frames, rig=None, camera_type="pinhole"):
        """frames: sequence (length T) of per-view lists of `Frame` (a single `Frame` per step for one view)."""
        frames = [f if isinstance(f, (list, tuple)) else [f] for f in frames]
        total, n_views = len(frames), len(frames[0])
        assert total > 0 and all(len(f) == n_views for f in frames)
        if rig is None:
            assert n_views == 1, "Need rig for multiple views"
            rig = SE3.Identity(1)
        height, width = frames[0][0].rgb.shape[:2]
        self.config.frontend.has_init_pose = frames[0][0].pose is not None
        self._build_components(height, width, n_views, rig, camera_type)
        b = self.buffer
        import gc
        gc_was = self.config.pause_gc and gc.isenabled()
        if gc_was:
            gc.collect()
            gc.disable()
        try:
            self._run_passes(frames, total)
        finally:
            if gc_was:
                gc.enable()
        filled = self.inner_filler.get_result()
        if filled.poses.data.shape[0] != total:
            raise ValueError("fewer poses than frames: the frame sequence changed between the passes")
        slam_map = b.extract_slam_map(filter_thresh=self.config.map_filter_thresh)
        return SLAMOutput(trajectory=filled.poses.inv(), intrinsics=b.intrinsics.clone(), rig=SE3(b.rig.clone()),
                          slam_map=slam_map)
