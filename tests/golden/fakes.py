"""Recording stand-ins for `FactorGraph` and `GraphBuffer`, shared by tests/golden/make_golden.py (which drives the
REFERENCE's SLAMFrontend / SLAMBackend / InnerFiller with them and freezes the call traces) and tests/test_oracle_golden.py
(which drives this repository's classes with them and compares).  Test infrastructure: nothing numeric of the hot path runs
here - the fakes evolve just enough state (edge ages, keyframe count, poses) for the schedulers' control flow."""
import numpy as np
import torch


def _lst(x):
    if x is None:
        return None
    if torch.is_tensor(x):
        return x.detach().cpu().reshape(-1).tolist()
    return np.asarray(x).reshape(-1).tolist()


class FakeBuffer:
    """what the schedulers read / write of `GraphBuffer`: frame count, poses, disparities, time stamps, the dirty flags, and
    the frame distance of the keyframe test (scripted values, popped per call)"""

    def __init__(self, n_frames, n_views=1, size=40, ht=3, wd=4, seed=0, distances=()):
        g = torch.Generator().manual_seed(seed)
        self.n_frames, self.n_views, self.device = n_frames, n_views, torch.device("cpu")
        self.trace = []
        q = torch.randn(size, 4, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 0.0, 1.0])
        self.poses = torch.cat([torch.randn(size, 3, generator=g) * 0.3, q / q.norm(dim=-1, keepdim=True)], -1).float()
        self.disps = (0.2 + torch.rand(size, n_views, ht, wd, generator=g)).float()
        self.disps_sens = torch.zeros(size, n_views, ht, wd)
        self.tstamp = torch.zeros(size, dtype=torch.long)
        self.dirty = torch.zeros(size, dtype=torch.bool)
        self.distances = list(distances)
        self.geom_version = 0

    def touch(self):
        self.geom_version += 1

    def frame_distance_dense_disp(self, ii, jj, beta=0.3, bidirectional=True, **kw):
        self.trace.append(("frame_distance", _lst(ii), _lst(jj), float(beta), bool(bidirectional)))
        d = self.distances.pop(0)
        return torch.full((len(_lst(ii)), self.n_views), float(d))

    def update_disps_sens(self, depth_model, frame_idx=None):
        self.trace.append(("update_disps_sens", frame_idx))

    def build_adaptive_cross_view_idx(self):
        self.trace.append(("build_adaptive_cross_view_idx",))

    def remove_second_newest(self, ix):
        for a in (self.poses, self.disps, self.tstamp):
            a[ix] = a[ix + 1]
        self.n_frames -= 1


class FakeGraph:
    """records every call the schedulers make on a `FactorGraph`; edges are (i, j, age) rows: `add_*` appends a scripted
    number of edges ending at the newest keyframe, `update` ages them, `rm_factors` drops the masked ones"""
    instances = []
    edges_per_add = 3

    def __init__(self, net, video, device, max_factors=48, incremental=True, cross_view=False):
        self.video, self.trace = video, video.trace
        self.trace.append(("FactorGraph", int(max_factors), bool(incremental), bool(cross_view)))
        self.corr = None
        self._e = np.zeros((0, 3), dtype=np.int64)
        FakeGraph.instances.append(self)

    # -- the reference's attributes ... and this repository's host mirror
    @property
    def ii(self):
        return torch.from_numpy(self._e[:, 0].copy())

    @property
    def jj(self):
        return torch.from_numpy(self._e[:, 1].copy())

    @property
    def age(self):
        return torch.from_numpy(self._e[:, 2].copy())

    def host_edges(self):
        return {"ii": self._e[:, 0].copy(), "jj": self._e[:, 1].copy(), "age": self._e[:, 2].copy(),
                "ii_inac": np.zeros(0, dtype=np.int64), "jj_inac": np.zeros(0, dtype=np.int64)}

    def _append(self, k):
        t = self.video.n_frames - 1
        new = np.array([[max(t - 1 - a, 0), t, 0] for a in range(k)], dtype=np.int64).reshape(-1, 3)
        self._e = np.concatenate([self._e, new], 0)
        self.corr = object()

    def add_factors(self, ii, jj, remove=False):
        self.trace.append(("add_factors", _lst(ii), _lst(jj), bool(remove)))
        new = np.stack([np.asarray(_lst(ii)), np.asarray(_lst(jj)), np.zeros(len(_lst(ii)))], 1).astype(np.int64)
        self._e = np.concatenate([self._e, new], 0)
        self.corr = object()

    def add_neighborhood_factors(self, t0, t1, r=3):
        self.trace.append(("add_neighborhood_factors", int(t0), int(t1), int(r)))
        self._append(self.edges_per_add)

    def add_proximity_factors(self, t0=0, t1=0, rad=2, nms=2, beta=0.25, thresh=16.0, remove=False, **kw):
        self.trace.append(("add_proximity_factors", int(t0), int(t1), int(rad), int(nms), float(beta), float(thresh), bool(remove)))
        self._append(self.edges_per_add)

    def rm_factors(self, mask, store=False):
        m = np.asarray(_lst(mask), dtype=bool)
        self.trace.append(("rm_factors", m.tolist(), bool(store)))
        self._e = self._e[~m]

    def rm_second_newest_keyframe(self, ix):
        self.trace.append(("rm_second_newest_keyframe", int(ix)))
        self.video.remove_second_newest(ix)
        self._e = self._e[(self._e[:, 0] != ix) & (self._e[:, 1] != ix)]
        self._e[:, :2] -= (self._e[:, :2] >= ix)

    def update(self, t0=None, t1=None, itrs=3, use_inactive=False, motion_only=False, fixed_motion=False, limited_disp=False):
        self.trace.append(("update", t0, t1, int(itrs), bool(use_inactive), bool(motion_only), bool(fixed_motion), bool(limited_disp)))
        self._e[:, 2] += 1

    def update_batch(self, itrs, steps, optimize_intrinsics, optimize_rig_rotation, solver_verbose=False):
        self.trace.append(("update_batch", int(itrs), int(steps), bool(optimize_intrinsics), bool(optimize_rig_rotation)))


def frontend_scenario():
    """(warmup keyframes, number of further keyframes, scripted keyframe distances - the 2nd and 5th are below the 4.0
    threshold: those keyframes are dropped again)"""
    return 8, 9, [5.0, 3.0, 6.0, 7.5, 1.0, 9.0, 4.5, 5.5, 8.0]


def run_frontend(cls, args, has_init_pose):
    """drive a SLAMFrontend class (the reference's or this repository's) through the scenario -> (trace, poses, disps)"""
    warm, more, dist = frontend_scenario()
    video = FakeBuffer(0, distances=dist)
    FakeGraph.instances.clear()
    fe = cls(None, video, args, torch.device("cpu"))
    for _ in range(warm + more):
        video.n_frames += 1
        fe.run()
    return video.trace, video.poses.clone(), video.disps.clone(), int(fe.t1), int(video.n_frames)
