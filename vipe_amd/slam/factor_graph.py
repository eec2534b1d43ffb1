"""Factor graph: edge lists + the update iteration (host-side mirror of vipe/slam/components/factor_graph.py).

`update()` is the hot loop A of SURVEY.md section 3.3 (factor_graph.py:231-314): reproject -> 4-level
correlation lookup -> flow-update operator -> dense BA.  Per iteration it issues 2 + (GRU convolutions) +
1 C-ABI calls on the current stream and never synchronises the host (the reference syncs in torch.unique,
scatter_mean's index.max(), every _tmult_mat_elements and the CPU spsolve).
"""

import os

import numpy as np
import torch

from .._lib import upload, upload_many
from ..ext import slam_ext
from .networks import AltCorrBlock, CorrBlock, CorrPool

# host-side work counters (edge counts known without touching the device): edges x operator applications, correlation
# pyramids built - what bench.py prices a whole clip's algorithmic bytes / FLOPs with (SURVEY 8d per-edge figures)
WORK = {"edge_updates": 0, "pyramids_built": 0}


class _Growable:
    """Per-edge state with spare capacity: `append` writes the new rows behind the live ones (the reference concatenates,
    i.e. copies the whole tensor, on every `add_factors`: factor_graph.py:160-170 - 4.4 MB per edge for the operator's
    input buffer and the hoisted gate context), `select` gathers the surviving rows into the OTHER of two backing buffers.
    `view` is what the rest of the code sees: an ordinary contiguous tensor [n, ...]."""

    def __init__(self):
        self.bufs = [None, None]
        self.cur = 0
        self.n = 0

    @property
    def view(self):
        return None if self.bufs[self.cur] is None else self.bufs[self.cur][:self.n]

    def _room(self, which, rows, like):
        b = self.bufs[which]
        if b is None or b.shape[0] < rows or b.shape[1:] != like.shape[1:] or b.dtype != like.dtype or b.device != like.device:
            # 25 % spare: the frontend's window oscillates by a few edges; the backend's graphs (tens of GB) must not double
            nb = torch.empty((max(rows + rows // 4, 64),) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
            if b is not None and which == self.cur and self.n:
                nb[:self.n] = b[:self.n]
            self.bufs[which] = nb
        return self.bufs[which]

    def append(self, x):
        b = self._room(self.cur, self.n + x.shape[0], x)
        b[self.n:self.n + x.shape[0]] = x
        self.n += x.shape[0]
        return self.view

    def select(self, idx):
        src = self.view
        other = self.cur ^ 1
        dst = self._room(other, max(int(idx.shape[0]), 1), src)
        if idx.shape[0]:
            torch.index_select(src, 0, idx, out=dst[:idx.shape[0]])
        self.cur, self.n = other, int(idx.shape[0])
        return self.view


class FactorGraph:
    def __init__(self, update_module, buffer, device, max_factors=48, incremental=True, cross_view=False):
        self.update_op = update_module
        self.buffer = buffer
        self.device = device
        self.max_factors = max_factors
        self.incremental = incremental
        self.cross_view = cross_view and buffer.n_views > 1  # factor_graph.py:62
        ht, wd = buffer.height // 8, buffer.width // 8
        self.ht, self.wd = ht, wd
        self.ii = torch.as_tensor([], dtype=torch.long, device=device)
        self.jj = torch.as_tensor([], dtype=torch.long, device=device)
        self._age = torch.as_tensor([], dtype=torch.long, device=device)
        self._age_lag = 0
        self.damping = 1e-6 * torch.ones_like(buffer.flattened_disps)  # factor_graph.py:76
        self.target = torch.zeros([1, 0, ht, wd, 2], device=device, dtype=torch.float)
        self.weight = torch.zeros([1, 0, ht, wd, 2], device=device, dtype=torch.float)
        # channels-last state of the flow-update operator: hidden state [E,h,w,128] and [inp | corr | flow] features
        self.corr, self.net_n, self.xbuf = None, None, None
        self.pgate = None  # [E,h,w,384]: context-feature part of the GRU gates, computed once per edge
        self._xbuf_store, self._pgate_store = _Growable(), _Growable()  # backing stores of xbuf / pgate (spare capacity)
        # hidden-state part of the gates for the CURRENT net_n (UpdateEngine.hidden_gate_state), computed on a second
        # stream in the shadow of the previous iteration's BA; None whenever net_n / the edge set changed since
        self._gate_state = None
        self._side = None
        # tuning of that stage ("the BA's shadow"): taken from this many active edges on; released by
        # the BA per Gauss-Newton iteration ("gated") or launched at once ("free"); share of the edges whose z|r part is
        # staged; shares of the pieces
        self.gate_overlap_min_edges = 64
        self.gate_overlap_mode = "gated"
        self.gate_overlap_share = 0.5
        self.gate_overlap_fractions = [0.4, 0.4, 0.2]
        self.ii_inac = torch.as_tensor([], dtype=torch.long, device=device)
        self.jj_inac = torch.as_tensor([], dtype=torch.long, device=device)
        self.target_inac = torch.zeros([1, 0, ht, wd, 2], device=device, dtype=torch.float)
        self.weight_inac = torch.zeros([1, 0, ht, wd, 2], device=device, dtype=torch.float)
        self._plan = None
        self._plan_serial = 0   # counts rebuilt edge plans: (serial, kind) identifies the index arrays handed to the BA
        self._ba_state = {}     # private BA workspace + the key of the plan it holds (slam_ext.dense_ba)
        self._h = None  # host mirror of the integer edge state (ii, jj, age, ii_inac, jj_inac), see host_edges()

    @property
    def age(self):
        """factor_graph.py:72 - edge ages on the device.  `update` only counts its calls (the host mirror is what the
        scheduling logic reads); the device tensor catches up when somebody looks at it."""
        if self._age_lag:
            self._age += self._age_lag
            self._age_lag = 0
        return self._age

    @age.setter
    def age(self, value):
        self._age, self._age_lag = value, 0

    def host_edges(self):
        """Host-side copy of the integer edge bookkeeping {ii, jj, age, ii_inac, jj_inac} (numpy int64).  Every method
        of this class that changes the edge lists updates the mirror alongside the device tensors, so the scheduling
        logic around the update iteration (duplicate filtering, suppression, age eviction, plan building) reads no
        device memory back.  If the tensors were replaced from outside, the mirror is rebuilt from them (one read-back)."""
        h = getattr(self, "_h", None)
        age = getattr(self, "age", None)
        if (h is None or h["ii"].shape[0] != self.ii.shape[0] or h["ii_inac"].shape[0] != self.ii_inac.shape[0]
                or (age is not None and h["age"].shape[0] != age.shape[0])):
            h = {k: getattr(self, k).detach().cpu().numpy().astype(np.int64).copy() for k in ("ii", "jj", "ii_inac", "jj_inac")}
            h["age"] = age.detach().cpu().numpy().astype(np.int64).copy() if age is not None else np.zeros_like(h["ii"])
            self._h = h
        return h

    def _filter_repeated_edges(self, ii, jj):
        """factor_graph.py:96-108 on the host mirror (the reference pays a .item() per edge).  ii, jj: numpy."""
        have = self._edge_set()
        keep = np.array([(i, j) not in have for i, j in zip(ii.tolist(), jj.tolist())], dtype=bool)
        return ii[keep], jj[keep]

    def _edge_set(self):
        """{(i, j)} of all active + inactive edges, kept incrementally (the inactive list grows with the video: rebuilding
        the set on every call is O(length of the video) of Python per keyframe).  Rebuilt when it is out of step."""
        h = self.host_edges()
        n = h["ii"].shape[0] + h["ii_inac"].shape[0]
        have = getattr(self, "_have", None)
        if have is None or len(have) != n:
            have = set(zip(h["ii"].tolist(), h["jj"].tolist())) | set(zip(h["ii_inac"].tolist(), h["jj_inac"].tolist()))
            self._have = have
        return have

    @torch.no_grad()
    def add_factors(self, ii, jj, remove=False):
        """factor_graph.py:119-173."""
        ii_h = (ii.detach().cpu().numpy() if torch.is_tensor(ii) else np.asarray(ii)).astype(np.int64).reshape(-1)
        jj_h = (jj.detach().cpu().numpy() if torch.is_tensor(jj) else np.asarray(jj)).astype(np.int64).reshape(-1)
        ii_h, jj_h = self._filter_repeated_edges(ii_h, jj_h)
        if ii_h.shape[0] == 0:
            return
        if (self.max_factors > 0 and self.ii.shape[0] + ii_h.shape[0] > self.max_factors and self.corr is not None
                and remove):
            # factor_graph.py:136-139 (the reference's `arange[argsort(age)]` is the permutation itself; torch's sort
            # leaves the order of equal ages unspecified - the stable order is used here)
            ix = np.argsort(self.host_edges()["age"], kind="stable")
            self.rm_factors(ix >= self.max_factors - ii_h.shape[0], store=True)
        ii, jj = upload_many([ii_h, jj_h], self.device)
        pi, qi, _, pj, qj, _ = self.buffer.expand_edge_multiview(ii, jj)
        if self.incremental:
            if self.corr is None:
                self.corr = CorrPool(capacity=max(64, self.max_factors + 16))
            V = self.buffer.n_views
            lo, hi = int(min(ii_h.min(), jj_h.min())), int(max(ii_h.max(), jj_h.max()))
            f1, f2 = (pi, pj) if V == 1 else (pi * V + qi, pj * V + qj)  # one view: frame index = pose index
            # a cross-view self edge (i, i) is re-targeted to cross_view_idx[i, v] by expand_edge_multiview - after an
            # adaptive cross-view pass that may be ANY keyframe of the buffer: the host (ii, jj) do not bound the frames
            rng = (0, self.buffer.n_frames * V) if (self.cross_view and V > 1) else (lo * V, (hi + 1) * V)
            self.corr.add_edges(self.buffer.flattened_fmaps, f1, f2, frame_range=rng)
            WORK["pyramids_built"] += int(pi.shape[0])
            xb = torch.zeros((ii.shape[0] * self.buffer.n_views, self.ht, self.wd, 320), dtype=torch.half,
                             device=self.device)
            xb[..., 0:128] = self.buffer.inps[pi, qi].permute(0, 2, 3, 1)
            self.xbuf = self._xbuf_store.append(xb)
            eng = self.update_op.engine(self.device)
            if eng.supports_gate_split(self.ht, self.wd):
                self.pgate = self._pgate_store.append(eng.gate_context(xb))
        target, _ = self.buffer.reproject_dense_disp(ii, jj)
        target = target[None]
        h = self.host_edges()
        self._edge_set().update(zip(ii_h.tolist(), jj_h.tolist()))
        h["ii"], h["jj"] = np.concatenate([h["ii"], ii_h]), np.concatenate([h["jj"], jj_h])
        h["age"] = np.concatenate([h["age"], np.zeros_like(ii_h)])
        self.ii = torch.cat([self.ii, ii], 0)
        self.jj = torch.cat([self.jj, jj], 0)
        self.age = torch.cat([self.age, torch.zeros_like(ii)], 0)
        net = self.buffer.nets[pi, qi].permute(0, 2, 3, 1).contiguous()
        self.net_n = net if self.net_n is None else torch.cat([self.net_n, net], 0)
        self._gate_state = None
        self.target = torch.cat([self.target, target], 1)
        self.weight = torch.cat([self.weight, torch.zeros_like(target)], 1)
        self._plan = None

    @torch.no_grad()
    def rm_factors(self, mask, store=False):
        """factor_graph.py:175-202.  The mask is read back ONCE; every tensor is then compacted with the same index
        vectors (boolean-mask indexing would synchronise per tensor)."""
        m = (mask.detach().cpu().numpy() if torch.is_tensor(mask) else np.asarray(mask)).astype(bool)
        h = self.host_edges()
        if store:
            h["ii_inac"] = np.concatenate([h["ii_inac"], h["ii"][m]])
            h["jj_inac"] = np.concatenate([h["jj_inac"], h["jj"][m]])
        elif getattr(self, "_have", None) is not None:
            self._have.difference_update(zip(h["ii"][m].tolist(), h["jj"][m].tolist()))
        h["ii"], h["jj"], h["age"] = h["ii"][~m], h["jj"][~m], h["age"][~m]
        V = self.buffer.n_views
        keep, drop = upload_many([np.flatnonzero(~m), np.flatnonzero(m)], self.device)
        views = torch.arange(V, device=self.device)
        keep_x = (keep[:, None] * V + views).view(-1) if V > 1 else keep
        drop_x = (drop[:, None] * V + views).view(-1) if V > 1 else drop
        keep_np = np.flatnonzero(~m)
        keep_x_np = (keep_np[:, None] * V + np.arange(V)).reshape(-1) if V > 1 else keep_np
        if store:
            self.ii_inac = torch.cat([self.ii_inac, self.ii[drop]], 0)
            self.jj_inac = torch.cat([self.jj_inac, self.jj[drop]], 0)
            self._append_inactive(self.target[:, drop_x], self.weight[:, drop_x])
        self.ii, self.jj, self.age = self.ii[keep], self.jj[keep], self.age[keep]
        if self.corr is not None:
            self.corr = self.corr[keep_x_np]  # host-side index: the pool only edits its slot vector
        self._gate_state = None
        if self.net_n is not None:
            self.net_n = self.net_n[keep_x]
        if self.xbuf is not None:
            self.xbuf = self._xbuf_store.select(keep_x)
        if self.pgate is not None:
            self.pgate = self._pgate_store.select(keep_x)
        self.target = self.target[:, keep_x]
        self.weight = self.weight[:, keep_x]
        self._plan = None

    def add_neighborhood_factors(self, t0, t1, r=3):
        """factor_graph.py:396-409 (mono: c = 0)."""
        ii, jj = torch.meshgrid(torch.arange(t0, t1), torch.arange(t0, t1), indexing="ij")
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        keep = ((ii - jj).abs() > 0) & ((ii - jj).abs() <= r)
        self.add_factors(ii[keep], jj[keep])

    @torch.no_grad()
    def add_proximity_factors(self, t0=0, t1=0, rad=2, nms=2, beta=0.25, thresh=16.0, remove=False, dist=None):
        """Edge proposal (factor_graph.py:411-488): neighbourhood edges i-rad-1..i-1 <-> i for i in [t0, t), then
        proximity edges (i, j), i in [t0, t), j in [t1, t), j <= i - rad, in order of increasing frame distance with
        non-maximum suppression around every edge already present or just added; all made bidirectional.

        Integer-exact mirror of the reference: the frame distances come from ONE launch of the HIP `frame_distance`
        kernel and ONE device-to-host copy; the suppression / ordering logic (a few hundred candidates) then runs on
        the host exactly as the reference's Python does (which instead pays one `.item()` sync per candidate)."""
        assert t0 >= t1, "t0 should be a subset of t1"
        t = self.buffer.n_frames
        iin, jjn = np.meshgrid(np.arange(t0, t, dtype=np.int64), np.arange(t1, t, dtype=np.int64), indexing="ij")
        iin, jjn = iin.reshape(-1), jjn.reshape(-1)
        if iin.size == 0:
            return
        if dist is not None:  # frame distances of exactly these candidates, computed earlier (SLAMFrontend prefetch)
            d = np.array(dist, dtype=np.float32).reshape(-1)
            assert d.shape[0] == iin.shape[0]
        else:
            ii, jj = upload_many([iin, jjn], self.device)
            d = self.buffer.frame_distance_dense_disp(ii, jj, beta=beta).mean(-1).cpu().numpy().astype(np.float32)
        nj = t - t1

        def suppress(i, j):
            if (t0 <= i < t) and (t1 <= j < t):
                d[(i - t0) * nj + (j - t1)] = np.inf

        def suppress_nms(i, j):
            lim = max(min(abs(i - j) - 2, nms), 0)
            for di in range(-nms, nms + 1):
                for dj in range(-nms, nms + 1):
                    if abs(di) + abs(dj) <= lim:
                        suppress(i + di, j + dj)

        # edges already in the graph (active + inactive): the same suppression, all edges at once per window offset
        D = d.reshape(t - t0, nj)  # view of d
        h = self.host_edges()
        I = np.concatenate([h["ii"], h["ii_inac"]])
        J = np.concatenate([h["jj"], h["jj_inac"]])
        if I.size:
            lim = np.maximum(np.minimum(np.abs(I - J) - 2, nms), 0)
            for di in range(-nms, nms + 1):
                for dj in range(-nms, nms + 1):
                    sel = (abs(di) + abs(dj)) <= lim
                    a, b = I[sel] + di, J[sel] + dj
                    ok = (a >= t0) & (a < t) & (b >= t1) & (b < t)
                    D[a[ok] - t0, b[ok] - t1] = np.inf
        d[(iin - rad < jjn) | (d > thresh)] = np.inf
        es = []
        for i in range(t0, t):
            if self.cross_view:
                es.append((i, i))
                suppress(i, i)
            for j in range(max(i - rad - 1, 0), i):
                es.append((i, j))
                es.append((j, i))
                suppress(i, j)
        for k in np.argsort(d, kind="stable"):
            if d[k] > thresh:
                continue
            if len(es) > self.max_factors:
                break
            i, j = int(iin[k]), int(jjn[k])
            es.append((i, j))
            es.append((j, i))
            suppress_nms(i, j)
        if len(es) == 0:
            return
        e = np.asarray(es, dtype=np.int64)
        self.add_factors(e[:, 0], e[:, 1], remove)

    @torch.no_grad()
    def rm_second_newest_keyframe(self, ix):
        """factor_graph.py:204-228: drop keyframe ix (= n_frames - 2) from the buffer and the graph."""
        self.buffer.remove_second_newest(ix)
        self._have = None  # frame indices shift: the edge set is rebuilt on next use
        h = self.host_edges()
        m = (h["ii_inac"] == ix) | (h["jj_inac"] == ix)
        self.ii_inac = self.ii_inac - (self.ii_inac >= ix).long()
        self.jj_inac = self.jj_inac - (self.jj_inac >= ix).long()
        h["ii_inac"] = h["ii_inac"] - (h["ii_inac"] >= ix)
        h["jj_inac"] = h["jj_inac"] - (h["jj_inac"] >= ix)
        if m.any():
            V = self.buffer.n_views
            keep = upload(np.flatnonzero(~m), self.device)
            keep_x = (keep[:, None] * V + torch.arange(V, device=self.device)).view(-1) if V > 1 else keep
            self.ii_inac, self.jj_inac = self.ii_inac[keep], self.jj_inac[keep]
            self.target_inac = self.target_inac[:, keep_x]
            self.weight_inac = self.weight_inac[:, keep_x]
            h["ii_inac"], h["jj_inac"] = h["ii_inac"][~m], h["jj_inac"][~m]
        m = (h["ii"] == ix) | (h["jj"] == ix)
        self.ii = self.ii - (self.ii >= ix).long()
        self.jj = self.jj - (self.jj >= ix).long()
        h["ii"] = h["ii"] - (h["ii"] >= ix)
        h["jj"] = h["jj"] - (h["jj"] >= ix)
        self.rm_factors(m, store=False)

    def get_edges_np(self):
        """factor_graph.py:110-117."""
        ii, jj = self.ii.cpu().numpy(), self.jj.cpu().numpy()
        w = torch.mean(self.weight, dim=[0, 2, 3, 4]).cpu().numpy()
        ix = np.argsort(ii)
        return np.stack([ii[ix], jj[ix]], axis=1), w[ix]

    @property
    def f_net(self):
        """GRU hidden state in the reference's layout [1,E,128,h,w] (factor_graph.py:84-86)."""
        return None if self.net_n is None else self.net_n.permute(0, 3, 1, 2)[None]

    @property
    def inp(self):
        return None if self.xbuf is None else self.xbuf[..., 0:128].permute(0, 3, 1, 2)[None]

    def _net_spare(self):
        """Ping-pong buffer for the new hidden state (the Q epilogue cannot write in place: 3x3 halo)."""
        sp = getattr(self, "_spare", None)
        if sp is None or sp.shape != self.net_n.shape or sp.data_ptr() == self.net_n.data_ptr():
            sp = torch.empty_like(self.net_n)
        self._spare = self.net_n  # the current state becomes next iteration's spare
        return sp

    def _edge_plan(self):
        """Index tensors that only change when the edge set changes (the reference recomputes them, with a
        torch.unique sync, on every update: factor_graph.py:267-268)."""
        if self._plan is None:
            pi, qi, di, pj, qj, _ = self.buffer.expand_edge_multiview(self.ii, self.jj)
            h = self.host_edges()
            V = self.buffer.n_views
            di_h = (h["ii"][:, None] * V + np.arange(V)).reshape(-1)  # = di for edges that are not cross-view self edges
            if self.cross_view or V > 1:
                du, dix = torch.unique(di, return_inverse=True)  # multi-view: take the expansion as computed on the device
                n_src = int(du.numel())
            else:
                du_h, dix_h = np.unique(di_h, return_inverse=True)
                du, dix = upload_many([du_h, dix_h], self.device)
                n_src = int(du_h.shape[0])
            from .update_engine import segment_csr
            self._plan_serial = getattr(self, "_plan_serial", 0) + 1
            if self.cross_view or V > 1:
                csr = segment_csr(dix, n_src)
            else:  # the same CSR from the host mirror (torch.bincount reads its maximum back: a stream drain)
                order_h = np.argsort(dix_h, kind="stable").astype(np.int32)
                rowptr_h = np.concatenate([[0], np.cumsum(np.bincount(dix_h, minlength=n_src))]).astype(np.int32)
                csr = tuple(upload_many([order_h, rowptr_h], self.device, torch.int32))
            self._plan = dict(pi=pi, qi=qi, di=di, pj=pj, qj=qj, du=du, dix=dix, n_src=n_src,
                              csr=csr,
                              t0=int(max(1, h["ii"].min() + 1)),
                              t1=int(max(h["ii"].max(), h["jj"].max()) + 1))
        return self._plan

    def _append_inactive(self, t_new, w_new):
        """target_inac / weight_inac grow for the whole video (factor_graph.py:184-189 concatenates, i.e. copies the
        whole store, on every eviction): append into buffers with spare capacity instead; the attributes stay views."""
        n, k = self.target_inac.shape[1], t_new.shape[1]
        cap = getattr(self, "_inac_cap", None)
        if cap is None or cap[0].shape[1] < n + k or cap[0].data_ptr() != self.target_inac.data_ptr():
            size = max(2 * (n + k), 256)
            cap = tuple(torch.empty((1, size) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
                        for x in (self.target_inac, self.weight_inac))
            cap[0][:, :n] = self.target_inac
            cap[1][:, :n] = self.weight_inac
            self._inac_cap = cap
        cap[0][:, n:n + k] = t_new
        cap[1][:, n:n + k] = w_new
        self.target_inac, self.weight_inac = cap[0][:, :n + k], cap[1][:, :n + k]

    def _shift_plan(self, plan5, base):
        """(pi, qi, di, pj, qj) -> the same relative to keyframe `base` (+ base), see GraphBuffer.bundle_adjustment"""
        pi, qi, di, pj, qj = plan5
        if self.cross_view or base <= 0:  # cross-view self edges may point anywhere in the buffer: keep absolute indices
            return (pi, qi, di, pj, qj, 0)
        V = self.buffer.n_views
        return (pi - base, qi, di - base * V, pj - base, qj, base)

    @torch.no_grad()
    def update(self, t0=None, t1=None, itrs=3, use_inactive=False, motion_only=False, fixed_motion=False,
               limited_disp=False):
        """run update operator on factor graph (factor_graph.py:230-314)."""
        assert self.incremental and self.corr is not None and self.xbuf is not None and self.net_n is not None
        assert not (motion_only and fixed_motion)
        P = self._edge_plan()
        t0 = P["t0"] if t0 is None else t0
        t1 = P["t1"] if t1 is None else t1
        buf = self.buffer
        E_act = int(P["pi"].shape[0])
        WORK["edge_updates"] += E_act
        if "io" not in P:  # per edge set: persistent coords / motion-feature buffers (stable addresses), frame masks
            P["io"] = (torch.empty((E_act, self.ht, self.wd, 2), dtype=torch.float32, device=self.device),
                       torch.empty((E_act, self.ht, self.wd, 4), dtype=torch.float16, device=self.device))
            P["mask"] = buf.masks[P["pi"], P["qi"]].contiguous()
        if not self.weight.is_contiguous() or not self.target.is_contiguous():
            self.weight, self.target = self.weight.contiguous(), self.target.contiguous()
        # motion features + coords1 in one launch (factor_graph.py:253-261)
        coords1, motn = slam_ext.reproject_motion_nhwc(buf.poses, buf.flattened_disps, buf.intrinsics, buf.rig,
                                                       P["pi"], P["qi"], P["pj"], P["qj"], P["di"],
                                                       self.target[0], camera=buf.camera_type, out=P["io"])
        eng = self.update_op.engine(self.device)
        # the lookup is deferred into the correlation encoder's first convolution (one kernel, no [E,h,w,200] tensor); the
        # whole operator is one natively sequenced library call
        corr = self.corr.lookup_deferred(coords1)
        self.net_n, dw, eta, _ = eng.forward_nhwc(self.net_n, self.xbuf, corr, motn, ix=P["dix"], n_src=P["n_src"],
                                                  net_out=self._net_spare(), csr=P["csr"], pgate=self.pgate,
                                                  gate_state=self._gate_state)
        self._gate_state = None
        # The new hidden state is final here, and the dense BA below keeps ONE workgroup busy for most of its time
        # (band Cholesky): everything of the next iteration's gates that depends on the hidden state alone - the
        # global-context terms and the hidden-state third of the z|r convolution, 18 % of the operator's FLOPs - is
        # issued on a second stream now and joins after the BA.  Speculative: wasted (off the critical path) when the
        # edge set changes before the next call.  Small edge sets are launch bound, not compute bound: not worth it.
        overlap = self.pgate is not None and E_act >= self.gate_overlap_min_edges
        ba_overlap = None
        if overlap:
            main = torch.cuda.current_stream(self.device)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            self._side.wait_stream(main)
            # the BA's shadow is shorter than the whole stage: the z|r part is staged for a share of the edges only (the
            # others keep the unsplit convolution), the small global-context part for all of them
            n_staged = max(1, int(round(self.gate_overlap_share * E_act)))
            if self.gate_overlap_mode == "gated":
                # handed to the BA, which releases it in one piece per Gauss-Newton iteration, each behind the start of
                # that iteration's solve (vipe_overlap_fn): the solve needs a whole CU's LDS and would otherwise queue
                # behind the convolution's workgroups
                with torch.cuda.stream(self._side):  # the small global-context part at once, next to the first accumulate
                    eng.hidden_gate_state(self.net_n, self.pgate, parts=1)
                # the last piece has nothing after its solve to hide behind: it is the smallest
                fr = self.gate_overlap_fractions if len(self.gate_overlap_fractions) == itrs else None
                gate_state, ov = eng.gate_state_job(self.net_n, self.pgate, fractions=fr, n_staged=n_staged)
                ba_overlap = ov(self._side)
            else:  # "free": launched at once, competes with every BA kernel
                with torch.cuda.stream(self._side):
                    gate_state = eng.hidden_gate_state(self.net_n, self.pgate, n_staged=n_staged)
        # factor_graph.py:270-276 in one launch: target = coords1 + delta, weight with masked frames zeroed
        # (`weight[:, masks[pi, qi]] = 0` without the host sync of a boolean-mask assignment), damping[du] = eta.
        # target / weight are rewritten IN PLACE: their addresses only change with the edge set, so consecutive
        # iterations chain through them and the steady state allocates nothing
        slam_ext.update_finish(coords1, dw, P["mask"], self.target, self.weight, eta, P["du"], self.damping)
        if use_inactive:
            # factor_graph.py:296-304.  The selection of inactive edges and its multiview expansion only change with
            # the edge sets (or t0): cached in the edge plan, so the steady-state iteration has no boolean-mask
            # indexing (a device-to-host sync each) and no re-expansion
            key = ("inac", t0)
            if key not in P:
                h = self.host_edges()
                sel_h = np.flatnonzero((h["ii_inac"] >= t0 - 3) & (h["jj_inac"] >= t0 - 3))  # host mirror: no read-back
                sel = upload(sel_h, self.device)
                ii = torch.cat([self.ii_inac[sel], self.ii], 0)
                jj = torch.cat([self.jj_inac[sel], self.jj], 0)
                V = buf.n_views
                sel_exp = sel if V == 1 else (sel.view(-1, 1) * V + torch.arange(V, device=self.device).view(1, -1)).view(-1)
                base = int(min(h["ii"].min(), h["jj"].min(), *(h[k][sel_h].min() for k in ("ii_inac", "jj_inac") if sel_h.size)))
                P[key] = (ii, jj, sel_exp, self._shift_plan(buf.expand_edge_multiview(ii, jj)[:5], base))
            ii, jj, sel_exp, plan = P[key]
            plan_key = (self._plan_serial, "inac", t0)
            # [selected inactive | active] targets / weights live in ONE buffer per edge set: the inactive part is
            # gathered once (it never changes), self.target / self.weight are views of the tail that the iteration
            # rewrites in place - the reference concatenates both every call (factor_graph.py:300-304)
            ckey = ("comb", t0)
            if ckey not in P:
                n_sel = int(sel_exp.shape[0])
                comb = tuple(torch.empty((1, n_sel + E_act) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
                             for x in (self.target, self.weight))
                comb[0][:, :n_sel] = self.target_inac.index_select(1, sel_exp)
                comb[1][:, :n_sel] = self.weight_inac.index_select(1, sel_exp)
                P[ckey] = (comb, n_sel)
            comb, n_sel = P[ckey]
            if self.target.data_ptr() != comb[0][:, n_sel:].data_ptr():
                comb[0][:, n_sel:] = self.target
                comb[1][:, n_sel:] = self.weight
                self.target, self.weight = comb[0][:, n_sel:], comb[1][:, n_sel:]
            target, weight = comb
        else:
            ii, jj, target, weight = self.ii, self.jj, self.target, self.weight
            if "ba_plan" not in P:  # cached expand_edge_multiview of the edge set, relative to the oldest keyframe used
                h = self.host_edges()
                P["ba_plan"] = self._shift_plan((P["pi"], P["qi"], P["di"], P["pj"], P["qj"]),
                                                int(min(h["ii"].min(), h["jj"].min())))
            plan = P["ba_plan"]
            plan_key = (self._plan_serial, "act")
        E = target.shape[1]
        buf.bundle_adjustment(target.view(E, -1, 2), weight.view(E, -1, 2), self.damping, ii, jj, t0,
                              t1 if not fixed_motion else t0, itrs, 1e-3, 0.1, motion_only, limited_disp, False, False,
                              plan=plan, ba_state=getattr(self, "_ba_state", None), plan_key=plan_key, overlap=ba_overlap)
        if overlap:
            main.wait_stream(self._side)
            self._gate_state = gate_state
        self._age_lag += 1
        if getattr(self, "_h", None) is not None and self._h["age"].shape[0] == self._age.shape[0]:
            self._h["age"] += 1

    @torch.no_grad()
    def update_batch(self, itrs, steps, optimize_intrinsics, optimize_rig_rotation, solver_verbose=False):
        """Backend update (hot loop B, factor_graph.py:316-394): volume-free correlation (AltCorrBlock over the
        buffer's feature maps), the flow-update operator applied in chunks of 8 source keyframes, then ONE dense BA
        over all edges with `itrs` Gauss-Newton iterations (t0 = 1, t1 = n_frames, pose damping 1e-5 / 1e-2)."""
        assert self.net_n is not None
        buf = self.buffer
        t = buf.n_frames
        eng = self.update_op.engine(self.device)
        # The reference's backend uses the volume-free AltCorrBlock to fit 24 GB devices (droid_net.py:121-176).  With
        # 288 GB of HBM the per-chunk volume (~33 MB per edge, ~1.5 GB per chunk of 8 source keyframes) is cheap, and
        # building it (one fused kernel, 10 us per edge) + the fused lookup is >10x faster than 49 x 128-channel dot
        # products per pixel and level; the volume kernels cover every grid (blocked store on the padded grid), AltCorrBlock
        # stays the reference-shaped alternative (VIPE_AMD_BACKEND_ALTCORR=1) for devices where the volumes do not fit.
        from ..ext import droid_net_ext
        use_volume = (droid_net_ext.fused_build_covers(buf.flattened_fmaps.shape[1], self.ht, self.wd, 4, buf.flattened_fmaps.dtype)
                      and os.environ.get("VIPE_AMD_BACKEND_ALTCORR") is None)
        corr_op = None if use_volume else AltCorrBlock(buf.flattened_fmaps[None])
        P = self._edge_plan()
        V = buf.n_views
        # The feature maps do not change during the `steps` passes: each chunk's correlation pyramid is built in the
        # first pass and kept for the others while all of them fit VIPE_AMD_BACKEND_VOLUME_GB (default 160 of 288 GB).
        G_, S_, R_ = droid_net_ext.blocked_dims(self.ht, self.wd)[:3]  # the store is padded to G x 64 sources, R x 4 rows, S x 32 columns
        vol_bytes = self.ii.shape[0] * V * (G_ * 64) * (R_ * 4 * S_ * 32) * 2 * (1 + 1 / 4 + 1 / 16 + 1 / 64)
        keep_vols = use_volume and steps > 1 and vol_bytes <= float(os.environ.get("VIPE_AMD_BACKEND_VOLUME_GB", "160")) * 2**30
        # The reference walks the source keyframes in groups of 8 (factor_graph.py:337-343) to bound memory.  The
        # operator couples edges only through GraphAgg's per-source-frame mean, so ANY partition that keeps every
        # source frame's edges together gives the same result: with 288 GB of HBM the groups are merged until a
        # chunk holds up to VIPE_AMD_BACKEND_CHUNK_EDGES edges (default 4096, ~110 GB of pyramids at 48 x 64) - normally one chunk.
        # Edge indices come from the host mirror; chunks are selected with index vectors, not masks.
        h_ = self.host_edges()
        ii_np, jj_np = h_["ii"], h_["jj"]
        assert jj_np.max() >= ii_np.max()
        E_all = ii_np.shape[0]
        # a chunk's pyramids must fit the volume budget (VIPE_AMD_BACKEND_VOLUME_GB) whatever the grid: 33 MB per edge at
        # 48 x 64, 238 MB at 64 x 128
        per_edge = vol_bytes / max(1, self.ii.shape[0] * V)
        max_edges = max(8, min(int(os.environ.get("VIPE_AMD_BACKEND_CHUNK_EDGES", "4096")),
                               int(float(os.environ.get("VIPE_AMD_BACKEND_VOLUME_GB", "160")) * 2**30 / per_edge)))
        cnt = np.bincount(ii_np)
        groups, cur, cur_n = [], [], 0
        for g0 in range(0, len(cnt), 8):  # the reference's groups of 8 source frames are the merge unit
            n8 = int(cnt[g0:g0 + 8].sum())
            if n8 == 0:
                continue
            if cur and cur_n + n8 > max_edges:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(g0)
            cur_n += n8
        if cur:
            groups.append(cur)
        # Everything about a chunk that does not change during the `steps` passes is prepared once: index vectors, the
        # source-node CSR, the frame masks, the context features in the operator's input buffer and - like the
        # frontend does per edge - the context-feature part of the GRU gates (19 % of the operator's FLOPs per pass)
        chunks = {}

        def chunk(gi):
            c = chunks.get(gi)
            if c is not None:
                return c
            sel = np.flatnonzero(np.isin(ii_np // 8 * 8, groups[gi]))
            c = dict(whole=sel.shape[0] == E_all)
            if c["whole"]:
                iis, jjs = self.ii, self.jj
                c["idx_x"] = None
            else:
                idx = upload(sel, self.device)
                c["idx_x"] = (idx[:, None] * V + torch.arange(V, device=self.device)).view(-1) if V > 1 else idx
                iis, jjs = self.ii[idx], self.jj[idx]
            pis, qis, dis, pjs, qjs, djs = buf.expand_edge_multiview(iis, jjs)
            dis_np = (ii_np[sel][:, None] * V + np.arange(V)).reshape(-1)
            du_np, dixs_np = np.unique(dis_np, return_inverse=True)
            du_d, dixs_d = upload_many([du_np, dixs_np], self.device)
            c.update(pis=pis, qis=qis, dis=dis, pjs=pjs, qjs=qjs, djs=djs, n=sel.shape[0] * V, n_src=int(du_np.shape[0]),
                     du=du_d, dixs=dixs_d, mask=buf.masks[pis, qis].contiguous())
            order_h = np.argsort(dixs_np, kind="stable").astype(np.int32)
            rowptr_h = np.concatenate([[0], np.cumsum(np.bincount(dixs_np, minlength=c["n_src"]))]).astype(np.int32)
            c["csr"] = tuple(upload_many([order_h, rowptr_h], self.device, torch.int32))
            xb = torch.empty((c["n"], self.ht, self.wd, 320), dtype=torch.half, device=self.device)
            xb[..., 0:128] = buf.inps[pis, qis].permute(0, 2, 3, 1)
            c["xb"] = xb
            c["pgate"] = eng.gate_context(xb) if eng.supports_gate_split(self.ht, self.wd) else None
            chunks[gi] = c
            return c

        vols = {}
        for _ in range(steps):
            coords1, motn = slam_ext.reproject_motion_nhwc(buf.poses, buf.flattened_disps, buf.intrinsics, buf.rig,
                                                           P["pi"], P["qi"], P["pj"], P["qj"], P["di"],
                                                           self.target[0].contiguous(), camera=buf.camera_type)
            for gi in range(len(groups)):
                c = chunk(gi)
                whole, idx_x, n = c["whole"], c["idx_x"], c["n"]
                take = (lambda x: x) if whole else (lambda x: x.index_select(0, idx_x))  # noqa: E731
                c1 = take(coords1)
                if use_volume:
                    vol = vols.get(gi)
                    if vol is None:
                        vol = CorrBlock.from_buffer(buf.flattened_fmaps, c["pis"] * V + c["qis"], c["pjs"] * V + c["qjs"],
                                                    frame_range=(0, t * V) if (self.cross_view and V > 1) else
                                                    (int(min(ii_np.min(), jj_np.min())) * V,
                                                     (int(max(ii_np.max(), jj_np.max())) + 1) * V))
                        WORK["pyramids_built"] += n
                        if keep_vols:
                            vols[gi] = vol
                    corr_n = vol.lookup_deferred(c1)
                else:
                    corr1 = corr_op(c1[None], c["dis"], c["djs"])  # [1,n,196,h,w] fp32
                    corr_n = torch.zeros((n, self.ht, self.wd, 200), dtype=torch.half, device=self.device)
                    corr_n[..., :196] = corr1[0].permute(0, 2, 3, 1)
                net, dw, eta, _ = eng.forward_nhwc(take(self.net_n).contiguous(), c["xb"], corr_n, take(motn).contiguous(),
                                                   ix=c["dixs"], n_src=c["n_src"], csr=c["csr"], pgate=c["pgate"])
                WORK["edge_updates"] += n
                self._gate_state = None
                if whole:
                    self.net_n = net if net.data_ptr() != self.net_n.data_ptr() else net.clone()
                    if not self.weight.is_contiguous() or not self.target.is_contiguous():
                        self.weight, self.target = self.weight.contiguous(), self.target.contiguous()
                    slam_ext.update_finish(c1, dw, c["mask"], self.target, self.weight, eta, c["du"], self.damping)
                else:
                    weight = dw[..., 2:4].masked_fill(c["mask"].unsqueeze(-1), 0.0)
                    self.net_n[idx_x] = net
                    self.target[0, idx_x] = c1 + dw[..., 0:2]
                    self.weight[0, idx_x] = weight
                    self.damping[c["du"]] = eta
            E = self.target.shape[1]
            buf.bundle_adjustment(self.target.view(E, -1, 2), self.weight.view(E, -1, 2), self.damping, self.ii, self.jj,
                                  1, t, itrs, 1e-5, 1e-2, False, False, optimize_intrinsics, optimize_rig_rotation,
                                  verbose=solver_verbose)
