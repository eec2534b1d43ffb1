"""Keyframe state + the geometry/BA entry points the factor graph calls.

Host-side mirror of vipe/slam/components/buffer.py for the hot path: same tensor names, shapes and dtypes
(buffer.py:79-176), same method signatures for `expand_edge_multiview` (:318-361), `bundle_adjustment`
(:373-390), `reproject_dense_disp` (:527) and `frame_distance_dense_disp` (:550).  The arithmetic is one
C-ABI call each (vipe_amd.ext.slam_ext); the Python Solver / term / sparse-block machinery of the
reference (vipe/slam/ba, vipe/slam/maths) has no counterpart here - it is what the fused kernels replace.
"""

from dataclasses import dataclass

import torch

from ..ext import slam_ext


@dataclass
class BAConfig:
    dense_disp_alpha: float = 0.001  # configs/slam/default.yaml:48-49


def fold_flow_terms(target, weight, target2, weight2):
    """Two `DenseDepthFlowTerm`s on the same edges (buffer.py:405-447: the dense flow and the sparse-track flow, both
    scaled by 0.001) share their Jacobians, so their normal equations J^T (W1 + W2) J and J^T (W1 r1 + W2 r2) are those of
    ONE term with weight W1 + W2 and the weighted mean of the two targets; the energies differ by a constant the solver
    never reads.  -> (target, weight) of that term for the fused BA."""
    w = weight + weight2
    t = (weight * target + weight2 * target2) / w.clamp(min=torch.finfo(w.dtype).tiny)
    return torch.where(w > 0, t, target).contiguous(), w.contiguous()


class GraphBuffer:
    def __init__(self, height, width, n_views=1, buffer_size=1024, init_disp=1.0, cross_view_idx=None,
                 ba_config=None, camera_type="pinhole", device=torch.device("cuda")):
        assert height % 8 == 0 and width % 8 == 0  # buffer.py:76
        if cross_view_idx is None:
            cross_view_idx = [(i + 1) % n_views for i in range(n_views)]
        self.n_frames = 0
        self.geom_version = 0  # counts writes to poses / disps / intrinsics made through this class, see touch()
        self.height, self.width, self.n_views, self.device = height, width, n_views, device
        self.ba_config = ba_config or BAConfig()
        self.camera_type = camera_type
        ht, wd = height // 8, width // 8
        f32 = dict(device=device, dtype=torch.float)
        self.tstamp = torch.zeros(buffer_size, device=device, dtype=torch.int)
        self.poses = torch.zeros(buffer_size, 7, **f32)
        self.poses[:, 6] = 1.0
        intr_dim = 5 if camera_type == "mei" else 4
        self.intrinsics = torch.zeros(n_views, intr_dim, **f32)
        self.rig = torch.zeros(n_views, 7, **f32)
        self.rig[:, 6] = 1.0
        self.disps = torch.ones(buffer_size, n_views, ht, wd, **f32) * init_disp
        self.disps_sens = torch.zeros(buffer_size, n_views, ht, wd, **f32)
        self.masks = torch.zeros(buffer_size, n_views, ht, wd, device=device, dtype=torch.bool)
        self._images = None  # full-resolution RGB 0-1 fp16 [N,V,3,H,W] (buffer.py:81-89), allocated on first use
        self.fmaps = torch.zeros(buffer_size, n_views, 128, ht, wd, device=device, dtype=torch.half)
        self.nets = torch.zeros(buffer_size, n_views, 128, ht, wd, device=device, dtype=torch.half)
        self.inps = torch.zeros(buffer_size, n_views, 128, ht, wd, device=device, dtype=torch.half)
        self.cross_view_idx = torch.zeros(buffer_size, n_views, 2, device=device, dtype=torch.long)
        self.cross_view_idx[..., 0] = torch.arange(buffer_size, device=device)[:, None]
        self.cross_view_idx[..., 1] = torch.tensor(cross_view_idx, device=device).long()[None]
        # buffer.py:106: frames changed since the last visualisation dump.  Only host code ever reads it: kept on the host
        # (the reference's device tensor costs a launch - and in the frontend an `ii.min()` read-back - per keyframe)
        self.dirty = torch.zeros(buffer_size, dtype=torch.bool)
        self.sparse_tracks = None  # buffer.py:73: a caller-supplied tracker (`enabled`, `compute_dense_disp_target_weight`); None = disabled
        self.last_depth_intrinsics = None  # intrinsics the sensor disparities were last estimated with (buffer.py:233-268)

    def touch(self):
        """Declare that poses / disps / intrinsics / rig changed.  Results derived from them and kept across calls (the
        frontend's prefetched frame distances) are only reused while this counter stands still; `bundle_adjustment` and
        `remove_second_newest` call it, code that writes the tensors directly must too."""
        self.geom_version += 1

    @property
    def images(self):
        if self._images is None:
            self._images = torch.zeros(self.tstamp.shape[0], self.n_views, 3, self.height, self.width, device=self.device,
                                       dtype=torch.float16)
        return self._images

    # ---- flattened (n v) views, buffer.py:181-199
    @property
    def flattened_disps(self):
        return self.disps.view(-1, *self.disps.shape[2:])

    @property
    def flattened_disps_sens(self):
        return self.disps_sens.view(-1, *self.disps_sens.shape[2:])

    @property
    def flattened_fmaps(self):
        return self.fmaps.view(-1, *self.fmaps.shape[2:])

    @property
    def flattened_nets(self):
        return self.nets.view(-1, *self.nets.shape[2:])

    @property
    def flattened_inps(self):
        return self.inps.view(-1, *self.inps.shape[2:])

    @property
    def K(self):
        """buffer.py:202-210: [V,3,3] pinhole calibration matrices at full resolution (numpy; MEI through its pinhole
        equivalent, cameras.py:338-343)."""
        import numpy as np
        intr = (self._pinhole_intrinsics_8() * 8.0).cpu().numpy()
        k_mat = np.eye(3)[None].repeat(self.n_views, axis=0)
        k_mat[:, 0, 0], k_mat[:, 1, 1], k_mat[:, 0, 2], k_mat[:, 1, 2] = intr[:, 0], intr[:, 1], intr[:, 2], intr[:, 3]
        return k_mat

    @property
    def K_dense_disp(self):
        """buffer.py:212-216: the same at the 1/8 resolution of the disparity maps."""
        k_mat = self.K
        k_mat[:, 0] /= 8
        k_mat[:, 1] /= 8
        return k_mat

    def expand_edge_multiview(self, ii, jj, cross=True, view_offset=0):
        """buffer.py:318-361 -> pi, qi, di, pj, qj, dj (each [M * n_views])."""
        V = self.n_views
        if V == 1:
            # one view: pose index = frame index, view index 0 - no device arithmetic at all (the general form below is
            # 14 small launches, three times per keyframe in the frontend).  The zeros are views of one cached buffer:
            # callers only read these index vectors
            ii_d, jj_d = ii.reshape(-1).to(self.device).contiguous(), jj.reshape(-1).to(self.device).contiguous()
            z = getattr(self, "_zeros_i64", None)
            n = max(int(ii_d.shape[0]), int(jj_d.shape[0]))
            if z is None or z.shape[0] < n or z.device != ii_d.device:
                z = self._zeros_i64 = torch.zeros(max(2 * n, 1024), dtype=torch.long, device=self.device)
            return ii_d, z[:ii_d.shape[0]], ii_d, jj_d, z[:jj_d.shape[0]], jj_d
        qi = torch.arange(V, device=self.device).reshape(1, -1).repeat(ii.shape[0], 1)
        pi = ii.reshape(-1, 1).repeat(1, V).to(self.device)
        qj = torch.arange(V, device=self.device).reshape(1, -1).repeat(jj.shape[0], 1)
        pj = jj.reshape(-1, 1).repeat(1, V).to(self.device)
        if cross and V > 1:
            cross_mask = ii == jj
            if torch.any(cross_mask):
                t, v = self.cross_view_idx[pi[cross_mask], qi[cross_mask]].unbind(-1)
                pj[cross_mask], qj[cross_mask] = t, v
        qj = (qj + view_offset) % V
        di = pi * V + qi
        dj = pj * V + qj
        return tuple(x.reshape(-1).contiguous() for x in (pi, qi, di, pj, qj, dj))

    def bundle_adjustment(self, target, weight, disp_damping, ii, jj, t0, t1, n_iters, pose_damping, pose_ep,
                          motion_only, limited_disp, optimize_intrinsics, optimize_rig_rotation, verbose=False,
                          plan=None, ba_state=None, plan_key=None, overlap=None):
        """buffer.py:373-525, in place on self.poses / self.disps (/ self.intrinsics).  `plan` = (pi, qi, di, pj, qj)
        of `expand_edge_multiview(ii, jj)` when the caller already holds it (the reference re-expands every call)."""
        assert t0 <= t1
        base = 0
        if plan is not None and len(plan) == 6:
            # (pi, qi, di, pj, qj, base): indices already relative to keyframe `base`, the oldest one any term touches.
            # The library then sees only the buffer rows from `base` on: its per-frame launches (grid.y = frames) do
            # not grow with the length of the video, only with the window the edges span.
            pi, qi, di, pj, qj, base = plan
        else:
            pi, qi, di, pj, qj = plan if plan is not None else self.expand_edge_multiview(ii, jj)[:5]
        V = self.n_views
        n_poses = max(self.n_frames, int(t1)) - base
        st = self.sparse_tracks
        if st is not None and getattr(st, "enabled", False):
            # buffer.py:422-447: a second flow term on the same edges, targets / weights from the caller's tracker
            assert V == 1, "This does not support cross-view tracking yet."
            s_target, s_weight = st.compute_dense_disp_target_weight(
                source_view_inds=qi, source_frame_inds=self.tstamp[pi + base], target_view_inds=qj,
                target_frame_inds=self.tstamp[pj + base], image_size=(self.height, self.width),
                dense_disp_size=(self.height // 8, self.width // 8))
            target, weight = fold_flow_terms(target, weight, s_target.flatten(1, 2), s_weight.flatten(1, 2))
        self.touch()
        return slam_ext.dense_ba(
            self.poses[base:], self.flattened_disps[base * V:], self.flattened_disps_sens[base * V:], self.intrinsics,
            self.rig, target.contiguous(), weight.contiguous(), disp_damping[base * V:].contiguous(), pi, qi, pj, qj, di,
            max(int(t0) - base, 0), max(int(t1) - base, 0), n_iters,
            pose_damping, pose_ep, motion_only, limited_disp, optimize_intrinsics, optimize_rig_rotation,
            camera=self.camera_type, alpha=self.ba_config.dense_disp_alpha, n_poses=n_poses, want_info=verbose,
            state=ba_state, plan_key=plan_key, overlap=overlap)

    def reproject_dense_disp(self, ii, jj):
        """buffer.py:527-548 -> coords [M,ht,wd,2], valid [M,ht,wd,1]."""
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        pi, qi, di, pj, qj, _ = self.expand_edge_multiview(ii, jj)
        return slam_ext.reproject(self.poses, self.flattened_disps, self.intrinsics, self.rig, pi, qi, pj, qj, di,
                                  camera=self.camera_type)

    def _pinhole_intrinsics_8(self):
        """camera_model.pinhole() at 1/8 scale (cameras.py:209-213, 338-348; geom.py:335)."""
        intr = self.intrinsics[:, :4] / 8.0
        if self.camera_type == "mei":
            intr = intr.clone()
            intr[:, 0:2] = intr[:, 0:2] / (1 + self.intrinsics[:, 4:5])
        return intr.contiguous()

    def remove_second_newest(self, ix):
        """buffer.py:218-231: keyframe ix is overwritten by its successor (the newest frame)."""
        assert ix == self.n_frames - 2
        self.touch()
        for name in ("tstamp", "_images", "poses", "disps", "disps_sens", "nets", "inps", "fmaps", "masks",
                     "cross_view_idx"):
            arr = getattr(self, name, None)
            if arr is not None:
                arr[ix] = arr[ix + 1]
        if getattr(self, "dirty", None) is not None:
            self.dirty[ix] = True
        self.n_frames -= 1

    def update_disps_sens(self, depth_model, frame_idx=None):
        """buffer.py:233-268: refresh the sensor-depth prior from a monocular depth model (one frame for the frontend, all
        frames for the backend after the intrinsics moved).  The model is the caller's (the reference's networks are
        outside the path): anything with `.depth_type` (str or enum whose value is "metric_depth" for models whose depth
        scales with the focal length) and `.estimate(inp) -> obj.metric_depth [V,H,W]`, where `inp` carries
        `rgb [V,H,W,3]` float 0-1 and `focal_length`."""
        from types import SimpleNamespace
        if depth_model is None:
            return
        if frame_idx is not None:
            frames = [int(frame_idx)]
        else:
            assert self.last_depth_intrinsics is not None
            if torch.allclose(self.last_depth_intrinsics, self.intrinsics):
                return
            dt = getattr(depth_model, "depth_type", None)
            if getattr(dt, "value", dt) == "metric_depth":  # depth already estimated: only the scale moves with the focal
                # (as the reference: `last_depth_intrinsics` is NOT advanced here, buffer.py:246-251 - a second call
                # rescales by the ratio to the focal of the original estimate again)
                self.disps_sens[: self.n_frames] *= self.last_depth_intrinsics[0][0].item() / self.intrinsics[0][0].item()
                return
            frames = range(self.n_frames)
        assert self.n_views == 1
        for f in frames:
            inp = SimpleNamespace(rgb=self.images[f].movedim(1, -1).float(), focal_length=self.intrinsics[0][0].item())
            disp = depth_model.estimate(inp).metric_depth[:, 3::8, 3::8]
            self.disps_sens[f] = torch.where(disp > 0, disp.reciprocal(), disp)
        self.last_depth_intrinsics = self.intrinsics.clone()

    def build_adaptive_cross_view_idx(self, valid_thresh=400.0):
        """buffer.py:270-301: for every (keyframe, view) the (keyframe, other view) with the smallest one-directional
        reprojection distance becomes its cross-view partner, where that distance is below `valid_thresh`."""
        if self.n_views == 1 or self.n_frames < 2:
            return
        n, V = self.n_frames, self.n_views
        ix = torch.arange(n, device=self.device)
        ii, jj = torch.meshgrid(ix, ix, indexing="ij")
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        ds = [self.frame_distance_dense_disp(ii, jj, beta=1.0, view_offset=off, bidirectional=False)
              .reshape(n, n, -1).permute(0, 2, 1) for off in range(1, V)]  # (source frame, view, target frame)
        d_total = torch.stack(ds, dim=-1).reshape(n, V, -1)
        d_min, inds_best = torch.min(d_total, dim=-1)
        t_best, off_best = inds_best // len(ds), inds_best % len(ds)
        tgt_view_best = (off_best + 1 + torch.arange(V, device=self.device)) % V
        new_inds = torch.stack([t_best, tgt_view_best], dim=-1)
        keep_old = ~(d_min < valid_thresh)
        new_inds[keep_old] = self.cross_view_idx[:n][keep_old]
        self.cross_view_idx[:n] = new_inds

    def frame_distance_dense_disp(self, ii, jj, beta=0.3, bidirectional=True, view_offset=0, n_frames=None, fused=True):
        """buffer.py:550-593 -> [M, n_views].  `n_frames`: frames the indices may address (default: the buffer's count).
        One launch (`vipe_frame_distance_rig`: the per-view poses R_v^-1 G_n of geom.py:338, the 1/8-scale pinhole
        intrinsics of geom.py:335, both directions and their mean inside the kernel); `fused=False` is the reference's
        sequence of operators - expanded poses for all frames, two `frame_distance` calls, the average - kept as the
        comparison form (same bits: tests/test_gpu_parity.py)."""
        from ..ext.lietorch import SE3

        pi, qi, di, pj, qj, dj = self.expand_edge_multiview(ii, jj, cross=False, view_offset=view_offset)
        V = self.n_views
        if fused and self.poses.is_cuda:
            d = slam_ext.frame_distance_rig(self.poses, self.rig, self.flattened_disps, self.intrinsics, pi, qi, pj, qj, beta,
                                            bidirectional)
            return d.view(-1, V)
        poses = SE3(self.poses[: self.n_frames if n_frames is None else n_frames])
        rig = SE3(self.rig)
        # expand poses into (n v) space: R_v^-1 * G_n   (geom.py:338)
        exp = (rig.inv().view((1, -1)) * poses.view((-1, 1))).view((-1,)).data.contiguous()
        intr = self._pinhole_intrinsics_8()
        # one view: frame index = pose index (no index arithmetic on the device: six launches per call otherwise)
        fi, fj = (pi, pj) if V == 1 else (pi * V + qi, pj * V + qj)
        d = slam_ext.frame_distance(exp, self.flattened_disps, intr, fi, fj, qi, qj, di, beta)
        if bidirectional:
            d2 = slam_ext.frame_distance(exp, self.flattened_disps, intr, fj, fi, qj, qi, dj, beta)
            d = 0.5 * (d + d2)
        return d.view(-1, V)

    def extract_slam_map(self, filter_thresh, t_range=None, is_local=False):
        """buffer.py:595-645: per keyframe and view the inverse-projected (world, or camera when `is_local`) points of
        the 1/8-resolution disparity map, the pixel colours at (3::8, 3::8), and the multi-view consistency mask
        (`depth_filter` count >= min(2, n-1), disparity above half the frame mean, not masked) -> SLAMMap."""
        from ..ext.lietorch import SE3
        from .interface import SLAMMap

        if t_range is None:
            t_range = torch.arange(self.n_frames, device=self.device)
        n = int(t_range.numel())
        c2w = SE3(self.poses[t_range]).inv()
        images = self.images[t_range][..., 3::8, 3::8].permute(0, 1, 3, 4, 2)
        pts_list, mask_list = [], []
        for v in range(self.n_views):
            c2w_view = c2w * SE3(self.rig[v])[None]
            disps_v = self.disps[t_range, v].contiguous()
            intr8 = self.intrinsics[v].clone()
            intr8[:4] = intr8[:4] / 8.0
            ident = SE3.Identity(n, device=self.device).data.contiguous()
            if self.camera_type == "pinhole":
                pts = slam_ext.iproj(ident if is_local else c2w_view.data.contiguous(), disps_v, intr8[:4].contiguous())
            else:  # MEI unprojection (cameras.py:228-250), then the same transform and dehomogenisation
                ht, wd = disps_v.shape[1:]
                yy, xx = torch.meshgrid(torch.arange(ht, device=self.device).float(),
                                        torch.arange(wd, device=self.device).float(), indexing="ij")
                fx, fy, cx, cy, k1 = intr8.unbind()
                ub, vb = (xx - cx) / fx, (yy - cy) / fy
                r2 = ub * ub + vb * vb
                fac = (k1 + torch.sqrt(1 + (1 - k1 * k1) * r2)) / (1 + r2)
                X, Y = ub * fac / (fac - k1), vb * fac / (fac - k1)
                p4 = torch.stack([X.expand(n, -1, -1), Y.expand(n, -1, -1), torch.ones_like(disps_v), disps_v], -1)
                if not is_local:
                    p4 = c2w_view[:, None, None].act(p4.contiguous())
                pts = p4[..., :3] / p4[..., 3:]
            thresh_v = filter_thresh * (1.0 / disps_v.mean().item())
            count = slam_ext.depth_filter(c2w_view.inv().data.contiguous(), disps_v,
                                          self._pinhole_intrinsics_8()[v].contiguous(),
                                          torch.arange(n, device=self.device),
                                          torch.full((n,), thresh_v, device=self.device))
            masks = ((count >= min(2, n - 1)) & (disps_v > 0.5 * disps_v.mean(dim=[1, 2], keepdim=True))
                     & (~self.masks[t_range, v]))
            pts_list.append(pts)
            mask_list.append(masks)
        return SLAMMap.from_masked_dense_disp(torch.stack(pts_list, 1), images, torch.stack(mask_list, 1),
                                              self.tstamp[t_range])
