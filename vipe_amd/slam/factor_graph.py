"""Factor graph: edge lists + the update iteration (host-side mirror of vipe/slam/components/factor_graph.py).

`update()` is the hot loop A of SURVEY.md section 3.3 (factor_graph.py:231-314): reproject -> 4-level
correlation lookup -> flow-update operator -> dense BA.  Per iteration it issues 2 + (GRU convolutions) +
1 C-ABI calls on the current stream and never synchronises the host (the reference syncs in torch.unique,
scatter_mean's index.max(), every _tmult_mat_elements and the CPU spsolve).
"""

import ctypes
import os

import numpy as np
import torch

from .._lib import check, lib, parse_struct, stream_ptr, upload, upload_many
from ..ext import slam_ext
from .networks import AltCorrBlock, CorrBlock, CorrPool

# host-side work counters (edge counts known without touching the device): edges x operator applications, correlation
# pyramids built - what bench.py prices a whole clip's algorithmic bytes / FLOPs with (SURVEY 8d per-edge figures)
WORK = {"edge_updates": 0, "pyramids_built": 0}


def warm_volume_pool(device, gigabytes=None):
    """Map the memory the first global BA of this process will ask for (`update_batch` keeps up to
    VIPE_AMD_BACKEND_VOLUME_GB of correlation pyramids) into torch's caching allocator NOW: a fresh process pays the
    driver ~0.9 s for the first 100 GB it maps (measured: the same 200-keyframe clip runs its two backend passes in
    1.7 s in a process that has held the memory before, 2.6 s in one that has not).  A long-lived worker that processes
    clip after clip is warm after its first clip; a benchmark's untimed warm-up should leave the process in that state."""
    if torch.device(device).type != "cuda":
        return 0
    want = float(gigabytes if gigabytes is not None else os.environ.get("VIPE_AMD_BACKEND_VOLUME_GB", "160")) * 2**30
    free, _ = torch.cuda.mem_get_info(device)
    n = int(min(want, 0.85 * free))
    if n <= 0:
        return 0
    x = torch.empty(n, dtype=torch.uint8, device=device)
    x[::1 << 21].zero_()  # touch every 2 MiB page
    del x
    return n


RowsJob = parse_struct("vipe_rows_job")   # include/vipe_amd.h
NhwcJob = parse_struct("vipe_nhwc_job")


class _EdgeStores:
    """The big per-edge tensors of an incremental graph - hidden state `net` [n,h,w,128], operator input `xbuf`
    [n,h,w,320] (context features in channels [0,128), the rest scratch), hoisted gate context `pgate` [n,h,w,384] - in
    backing buffers with spare capacity, two banks each.  The reference concatenates (copies the whole tensor) on every
    `add_factors` and compacts with boolean masks on every `rm_factors` (factor_graph.py:160-170, 190-201: 4.4 MB per
    edge); here `append` is ONE launch that reads the new edges' source frames from the keyframe buffer and writes them
    channels-last behind the live rows (`vipe_gather_nchw_to_nhwc_f16`), and `compact` is ONE launch that moves the
    surviving rows of all tensors (and whatever small per-edge arrays the caller adds) into the other bank
    (`vipe_rows_gather`).  The operator ping-pongs `net` between its two banks (`net_spare`)."""

    def __init__(self, ht, wd, device, with_pgate):
        self.ht, self.wd, self.device, self.with_pgate = ht, wd, device, with_pgate
        self.cap = 0
        self.net, self.xb, self.pg = [None, None], [None, None], [None, None]
        self.cur = 0  # bank of xbuf / pgate

    def _alloc(self, cap):
        f16 = dict(dtype=torch.float16, device=self.device)
        shp = (cap, self.ht, self.wd)
        return ([torch.empty(shp + (128,), **f16) for _ in range(2)], [torch.empty(shp + (320,), **f16) for _ in range(2)],
                [torch.empty(shp + (384,), **f16) if self.with_pgate else None for _ in range(2)])

    def bank_of(self, net_n):
        """which net bank `net_n` is a view of (None: a tensor from outside)"""
        if net_n is not None and self.net[0] is not None:
            for k in (0, 1):
                if net_n.data_ptr() == self.net[k].data_ptr():
                    return k
        return None

    def reserve(self, rows, net_n, n_live):
        """room for `rows` rows; -> net_n (re-homed when the stores grew or `net_n` came from outside)"""
        grow = rows > self.cap
        if grow:
            cap = max(rows + rows // 4, 64)  # 25 % spare: the frontend's window oscillates by a few edges
            net, xb, pg = self._alloc(cap)
            if n_live and self.cap:
                xb[0][:n_live, ..., :128] = self.xb[self.cur][:n_live, ..., :128]
                if self.with_pgate:
                    pg[0][:n_live] = self.pg[self.cur][:n_live]
            old = net_n
            self.net, self.xb, self.pg, self.cur, self.cap = net, xb, pg, 0, cap
            if n_live and old is not None:
                self.net[0][:n_live] = old
                return self.net[0][:n_live]
            return None
        if n_live and self.bank_of(net_n) is None:  # assigned from outside (tests restore a saved state): adopt it
            self.net[0][:n_live] = net_n
            return self.net[0][:n_live]
        return net_n

    def net_spare(self, net_n):
        k = self.bank_of(net_n)
        return None if k is None else self.net[1 - k][:net_n.shape[0]]

    def append(self, buffer_nets, buffer_inps, frames, net_n, n_live):
        """rows [n_live, n_live + k): hidden state and context features of the frames `frames` [k] int64 (device) of the
        keyframe buffer's flattened [N,128,h*w] maps -> (net_n, xbuf, new xbuf rows)"""
        k = int(frames.shape[0])
        if n_live == 0:
            net_n = None  # (a zero-row view has no address to recognise its bank by)
        net_n = self.reserve(n_live + k, net_n, n_live)
        nb = 0 if net_n is None else self.bank_of(net_n)
        P = self.ht * self.wd
        jobs = (NhwcJob * 2)()
        for j, (src, dst, ctot) in enumerate(((buffer_nets, self.net[nb], 128), (buffer_inps, self.xb[self.cur], 320))):
            jobs[j].src, jobs[j].frame, jobs[j].dst = src.data_ptr(), frames.data_ptr(), dst.data_ptr()
            jobs[j].dst_row_pitch, jobs[j].dst_ctot, jobs[j].dst_coff, jobs[j].dst_row0 = P * ctot, ctot, 0, n_live
        check(lib().vipe_gather_nchw_to_nhwc_f16(ctypes.addressof(jobs), 2, k, 128, P, stream_ptr(frames)), "gather_nchw_to_nhwc")
        n = n_live + k
        return self.net[nb][:n], self.xb[self.cur][:n], self.xb[self.cur][n_live:n]

    def compact(self, keep, n_keep, net_n, extra_jobs=()):
        """rows `keep` [n_keep] int64 (device, ascending) of every store -> the other bank, in ONE launch together with
        `extra_jobs` = (src, dst, idx, n_rows, row_bytes, dst_row0) of small per-edge arrays; -> (net_n, xbuf, pgate)"""
        P = self.ht * self.wd
        nb = self.bank_of(net_n)
        jobs = (RowsJob * 8)()
        n = 0

        def job(src, dst, idx, n_rows, row_bytes, dst_row0=0, seg=None):
            nonlocal n
            j = jobs[n]
            j.src, j.dst, j.idx = src.data_ptr(), dst.data_ptr(), (idx.data_ptr() if idx is not None else None)
            j.src_row_pitch = j.dst_row_pitch = row_bytes
            j.seg_bytes, j.seg_pitch, j.n_seg = seg if seg is not None else (row_bytes, row_bytes, 1)
            j.n_rows, j.dst_row0 = int(n_rows), int(dst_row0)
            n += 1

        if n_keep:
            job(self.net[nb], self.net[1 - nb], keep, n_keep, P * 128 * 2)
            job(self.xb[self.cur], self.xb[1 - self.cur], keep, n_keep, P * 320 * 2, seg=(256, 640, P))  # context features only
            if self.with_pgate:
                job(self.pg[self.cur], self.pg[1 - self.cur], keep, n_keep, P * 384 * 2)
        for e in extra_jobs:
            job(*e)
        check(lib().vipe_rows_gather(ctypes.addressof(jobs), n, stream_ptr(keep)), "rows_gather")
        self.cur ^= 1
        return (self.net[1 - nb][:n_keep], self.xb[self.cur][:n_keep], self.pg[self.cur][:n_keep] if self.with_pgate else None)


def _index_property(name):
    """Device copy of one integer array of the edge bookkeeping (`ii`, `jj`, `age`, `ii_inac`, `jj_inac`): the host mirror
    (`host_edges`) is what this class works on; the device tensor the reference keeps (factor_graph.py:66-75) is made from
    it when somebody reads the attribute, and an assignment from outside makes the mirror follow the tensor."""
    def get(self):
        dev = self.__dict__.setdefault("_dev", {})
        t = dev.get(name)
        if t is None:
            h = self.__dict__.get("_h")
            if h is None:
                return None
            t = dev[name] = upload(h[name], self.device)
        return t

    def put(self, value):
        self.__dict__.setdefault("_dev", {})[name] = value
        self.__dict__["_h"] = None  # replaced from outside: the mirror is rebuilt from the tensors on next use

    return property(get, put)


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
        # integer edge bookkeeping: the host mirror is authoritative, `ii` / `jj` / `age` / `ii_inac` / `jj_inac` are device
        # views of it made on demand (_index_property)
        z = np.zeros(0, dtype=np.int64)
        self._h = {"ii": z, "jj": z.copy(), "age": z.copy(), "ii_inac": z.copy(), "jj_inac": z.copy()}
        self._dev = {}
        self.damping = 1e-6 * torch.ones_like(buffer.flattened_disps)  # factor_graph.py:76
        self.target = torch.zeros([1, 0, ht, wd, 2], device=device, dtype=torch.float)
        self.weight = torch.zeros([1, 0, ht, wd, 2], device=device, dtype=torch.float)
        # channels-last state of the flow-update operator: hidden state [E,h,w,128] and [inp | corr | flow] features
        self.corr, self.net_n, self.xbuf = None, None, None
        self.pgate = None  # [E,h,w,384]: context-feature part of the GRU gates, computed once per edge
        self._stores = None  # _EdgeStores: backing buffers of net_n / xbuf / pgate (incremental graphs)
        # hidden-state part of the gates for the CURRENT net_n (UpdateEngine.hidden_gate_state), computed on a second
        # stream in the shadow of the previous iteration's BA; None whenever net_n / the edge set changed since
        self._gate_state = None
        self._side = None
        # tuning of that stage ("the BA's shadow"): taken from this many active edges on; released by
        # the BA per Gauss-Newton iteration ("gated") or launched at once ("free"); share of the edges whose z|r part is
        # staged; shares of the pieces
        self.gate_overlap_min_edges = int(os.environ.get("VIPE_AMD_GATE_OVERLAP_MIN_EDGES", "64"))
        self.gate_overlap_mode = "gated"
        self.gate_overlap_share = 0.5
        self.gate_overlap_fractions = [0.4, 0.4, 0.2]
        self.target_inac = torch.zeros([1, 0, ht, wd, 2], device=device, dtype=torch.float)
        self.weight_inac = torch.zeros([1, 0, ht, wd, 2], device=device, dtype=torch.float)
        self._plan = None
        self._plan_serial = 0   # counts rebuilt edge plans: (serial, kind) identifies the index arrays handed to the BA
        self._ba_state = {}     # private BA workspace + the key of the plan it holds (slam_ext.dense_ba)

    ii = _index_property("ii")
    jj = _index_property("jj")
    age = _index_property("age")  # factor_graph.py:72
    ii_inac = _index_property("ii_inac")
    jj_inac = _index_property("jj_inac")

    def _mirror_changed(self, *names):
        """the host mirror's arrays `names` were edited: their device copies are stale"""
        dev = self.__dict__.setdefault("_dev", {})
        for n in names:
            dev[n] = None

    def host_edges(self):
        """Host-side copy of the integer edge bookkeeping {ii, jj, age, ii_inac, jj_inac} (numpy int64) - what every method
        of this class works on, so that the scheduling logic around the update iteration (duplicate filtering, suppression,
        age eviction, plan building) reads no device memory back and issues no index arithmetic on the device.  If the
        attributes were assigned from outside, the mirror is rebuilt from those tensors (one read-back)."""
        h = self.__dict__.get("_h")
        if h is None:
            dev = self.__dict__.setdefault("_dev", {})
            h = {}
            for k in ("ii", "jj", "ii_inac", "jj_inac"):
                t = dev.get(k)
                h[k] = np.zeros(0, dtype=np.int64) if t is None else t.detach().cpu().numpy().astype(np.int64).copy()
            t = dev.get("age")
            h["age"] = (t.detach().cpu().numpy().astype(np.int64).copy() if t is not None and t.shape[0] == h["ii"].shape[0]
                        else np.zeros_like(h["ii"]))
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
        if (self.max_factors > 0 and self.host_edges()["ii"].shape[0] + ii_h.shape[0] > self.max_factors and self.corr is not None
                and remove):
            # factor_graph.py:136-139 (the reference's `arange[argsort(age)]` is the permutation itself; torch's sort
            # leaves the order of equal ages unspecified - the stable order is used here)
            ix = np.argsort(self.host_edges()["age"], kind="stable")
            self.rm_factors(ix >= self.max_factors - ii_h.shape[0], store=True)
        ii, jj = upload_many([ii_h, jj_h], self.device)
        pi, qi, _, pj, qj, _ = self.buffer.expand_edge_multiview(ii, jj)
        V = self.buffer.n_views
        n_live = 0 if self.net_n is None else int(self.net_n.shape[0])
        f1, f2 = (pi, pj) if V == 1 else (pi * V + qi, pj * V + qj)  # one view: frame index = pose index
        if self.incremental:
            if self.corr is None:
                self.corr = CorrPool(capacity=max(64, self.max_factors + 16))
            lo, hi = int(min(ii_h.min(), jj_h.min())), int(max(ii_h.max(), jj_h.max()))
            # a cross-view self edge (i, i) is re-targeted to cross_view_idx[i, v] by expand_edge_multiview - after an
            # adaptive cross-view pass that may be ANY keyframe of the buffer: the host (ii, jj) do not bound the frames
            rng = (0, self.buffer.n_frames * V) if (self.cross_view and V > 1) else (lo * V, (hi + 1) * V)
            self.corr.add_edges(self.buffer.flattened_fmaps, f1, f2, frame_range=rng)
            WORK["pyramids_built"] += int(pi.shape[0])
            eng = self.update_op.engine(self.device)
            if self._stores is None:
                self._stores = _EdgeStores(self.ht, self.wd, self.device, eng.supports_gate_split(self.ht, self.wd))
            # hidden state and context features of the new edges' source frames, straight from the keyframe buffer into
            # the stores' tails (one launch; the reference gathers, permutes and concatenates: factor_graph.py:147-170)
            self.net_n, self.xbuf, xb_new = self._stores.append(self.buffer.flattened_nets, self.buffer.flattened_inps,
                                                                f1.contiguous(), self.net_n, n_live)
            if self._stores.with_pgate:
                pg = self._stores.pg[self._stores.cur]
                eng.gate_context(xb_new, out=pg[n_live:n_live + xb_new.shape[0]])
                self.pgate = pg[:n_live + xb_new.shape[0]]
        else:
            net = self.buffer.nets[pi, qi].permute(0, 2, 3, 1).contiguous()
            self.net_n = net if self.net_n is None else torch.cat([self.net_n, net], 0)
        target, _ = self.buffer.reproject_dense_disp(ii, jj)
        target = target[None]
        h = self.host_edges()
        self._edge_set().update(zip(ii_h.tolist(), jj_h.tolist()))
        h["ii"], h["jj"] = np.concatenate([h["ii"], ii_h]), np.concatenate([h["jj"], jj_h])
        h["age"] = np.concatenate([h["age"], np.zeros_like(ii_h)])
        self._mirror_changed("ii", "jj", "age")
        self._gate_state = None
        self.target = torch.cat([self.target, target], 1)
        self.weight = torch.cat([self.weight, torch.zeros_like(target)], 1)
        self._plan = None

    @torch.no_grad()
    def rm_factors(self, mask, store=False):
        """factor_graph.py:175-202.  The mask is read ONCE (the callers of this package pass host arrays); the integer
        bookkeeping is edited on the host mirror, and the surviving rows of every per-edge tensor - hidden state, context
        features, gate context, targets, weights - and the rows that go to the inactive store move in ONE launch
        (`_EdgeStores.compact`; the reference indexes each tensor with the boolean mask, a sync per tensor)."""
        m = (mask.detach().cpu().numpy() if torch.is_tensor(mask) else np.asarray(mask)).astype(bool)
        if not m.any():
            return
        h = self.host_edges()
        if store:
            h["ii_inac"] = np.concatenate([h["ii_inac"], h["ii"][m]])
            h["jj_inac"] = np.concatenate([h["jj_inac"], h["jj"][m]])
            self._mirror_changed("ii_inac", "jj_inac")
        elif getattr(self, "_have", None) is not None:
            self._have.difference_update(zip(h["ii"][m].tolist(), h["jj"][m].tolist()))
        h["ii"], h["jj"], h["age"] = h["ii"][~m], h["jj"][~m], h["age"][~m]
        self._mirror_changed("ii", "jj", "age")
        V = self.buffer.n_views
        keep_np, drop_np = np.flatnonzero(~m), np.flatnonzero(m)
        keep_x_np = (keep_np[:, None] * V + np.arange(V)).reshape(-1) if V > 1 else keep_np
        drop_x_np = (drop_np[:, None] * V + np.arange(V)).reshape(-1) if V > 1 else drop_np
        keep_x, drop_x = upload_many([keep_x_np, drop_x_np], self.device)
        nk, nd = int(keep_x_np.shape[0]), int(drop_x_np.shape[0])
        if self.corr is not None:
            self.corr = self.corr[keep_x_np]  # host-side index: the pool only edits its slot vector
        self._gate_state = None
        self._plan = None
        if not self.target.is_contiguous() or not self.weight.is_contiguous():
            self.target, self.weight = self.target.contiguous(), self.weight.contiguous()
        row = int(self.target.shape[2] * self.target.shape[3] * 2 * 4)  # one edge's [h, w, 2] f32
        new_t = torch.empty((1, nk) + tuple(self.target.shape[2:]), dtype=self.target.dtype, device=self.device)
        new_w = torch.empty_like(new_t)
        extra = [(self.target, new_t, keep_x, nk, row, 0), (self.weight, new_w, keep_x, nk, row, 0)] if nk else []
        if store and nd:
            ti, wi, n0 = self._inactive_room(nd)
            extra += [(self.target, ti, drop_x, nd, row, n0), (self.weight, wi, drop_x, nd, row, n0)]
        st = self._stores
        if st is not None and self.net_n is not None and st.bank_of(self.net_n) is None and self.net_n.shape[0]:
            self.net_n = st.reserve(self.net_n.shape[0], self.net_n, self.net_n.shape[0])  # assigned from outside: adopt
        if st is not None and self.net_n is not None and st.bank_of(self.net_n) is not None:
            self.net_n, self.xbuf, pg = st.compact(keep_x, nk, self.net_n, extra)
            if self.pgate is not None:
                self.pgate = pg
        else:  # non-incremental graphs (the backend's): no stores, the small arrays in one launch all the same
            if self.net_n is not None:
                self.net_n = self.net_n[keep_x]
            if extra:
                jobs = (RowsJob * len(extra))()
                for j, (src, dst, idx, n_rows, rb, r0) in zip(jobs, extra):
                    j.src, j.dst, j.idx = src.data_ptr(), dst.data_ptr(), idx.data_ptr()
                    j.src_row_pitch = j.dst_row_pitch = j.seg_bytes = j.seg_pitch = rb
                    j.n_seg, j.n_rows, j.dst_row0 = 1, int(n_rows), int(r0)
                check(lib().vipe_rows_gather(ctypes.addressof(jobs), len(extra), stream_ptr(keep_x)), "rows_gather")
        self.target, self.weight = new_t, new_w

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
            d = self.buffer.frame_distance_dense_disp(ii, jj, beta=beta)
            d = (d[:, 0] if d.shape[1] == 1 else d.mean(-1)).cpu().numpy().astype(np.float32)
        nj = t - t1

        def suppress(i, j):
            if (t0 <= i < t) and (t1 <= j < t):
                d[(i - t0) * nj + (j - t1)] = np.inf

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
        # candidates in order of increasing distance; only those not above the threshold NOW can ever be taken (suppression
        # only raises distances): the global BA's 40 000 candidates shrink to a few thousand before the Python loop.  The
        # suppression diamonds are index offsets precomputed per radius (factor_graph.py:455-460 loops di, dj per edge)
        order = np.argsort(d, kind="stable")
        order = order[d[order] <= thresh]
        nt = t - t0
        diamonds = []
        for lim_k in range(nms + 1):
            dd = [(di, dj) for di in range(-nms, nms + 1) for dj in range(-nms, nms + 1) if abs(di) + abs(dj) <= lim_k]
            diamonds.append((np.array([x[0] for x in dd]), np.array([x[1] for x in dd])))
        for k in order.tolist():
            if d[k] > thresh:
                continue
            if len(es) > self.max_factors:
                break
            i, j = int(iin[k]), int(jjn[k])
            es.append((i, j))
            es.append((j, i))
            ddi, ddj = diamonds[max(min(abs(i - j) - 2, nms), 0)]
            a, b = i - t0 + ddi, j - t1 + ddj
            ok = (a >= 0) & (a < nt) & (b >= 0) & (b < nj)
            d[a[ok] * nj + b[ok]] = np.inf
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
        h["ii_inac"] = h["ii_inac"] - (h["ii_inac"] >= ix)
        h["jj_inac"] = h["jj_inac"] - (h["jj_inac"] >= ix)
        if m.any():
            V = self.buffer.n_views
            keep = upload(np.flatnonzero(~m), self.device)
            keep_x = (keep[:, None] * V + torch.arange(V, device=self.device)).view(-1) if V > 1 else keep
            self.target_inac = self.target_inac[:, keep_x]
            self.weight_inac = self.weight_inac[:, keep_x]
            h["ii_inac"], h["jj_inac"] = h["ii_inac"][~m], h["jj_inac"][~m]
        m = (h["ii"] == ix) | (h["jj"] == ix)
        h["ii"] = h["ii"] - (h["ii"] >= ix)
        h["jj"] = h["jj"] - (h["jj"] >= ix)
        self._mirror_changed("ii", "jj", "ii_inac", "jj_inac")
        self._plan = None
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
        """Ping-pong buffer for the new hidden state (the Q epilogue cannot write in place: 3x3 halo): the other bank of
        the edge stores, or - for graphs without stores and states assigned from outside - a private tensor."""
        sp = self._stores.net_spare(self.net_n) if self._stores is not None else None
        if sp is not None:
            return sp
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

    def _inactive_room(self, k):
        """target_inac / weight_inac grow for the whole video (factor_graph.py:184-189 concatenates, i.e. copies the
        whole store, on every eviction): backing buffers with spare capacity; -> (target buffer, weight buffer, first
        free row) with room for k more rows, the attributes re-pointed to the n + k rows (the caller fills the new ones)."""
        n = self.target_inac.shape[1]
        cap = getattr(self, "_inac_cap", None)
        if cap is None or cap[0].shape[1] < n + k or cap[0].data_ptr() != self.target_inac.data_ptr():
            size = max(2 * (n + k), 256)
            cap = tuple(torch.empty((1, size) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
                        for x in (self.target_inac, self.weight_inac))
            cap[0][:, :n] = self.target_inac
            cap[1][:, :n] = self.weight_inac
            self._inac_cap = cap
        self.target_inac, self.weight_inac = cap[0][:, :n + k], cap[1][:, :n + k]
        return cap[0], cap[1], n

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
                # [selected inactive | active] index vectors straight from the host mirror: one staged copy
                V = buf.n_views
                sel_x = sel_h if V == 1 else (sel_h[:, None] * V + np.arange(V)).reshape(-1)
                ii, jj, sel_exp = upload_many([np.concatenate([h["ii_inac"][sel_h], h["ii"]]),
                                               np.concatenate([h["jj_inac"][sel_h], h["jj"]]), sel_x], self.device)
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
        self.host_edges()["age"] += 1  # factor_graph.py:306
        self._mirror_changed("age")

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
        budget = float(os.environ.get("VIPE_AMD_BACKEND_VOLUME_GB", "160")) * 2**30
        keep_vols = use_volume and steps > 1 and vol_bytes <= budget
        # ... and when they do not (more than ~4800 edges at 48 x 64: clips of 300+ keyframes), as MANY chunks as fit next
        # to the one being worked on stay resident; the others are rebuilt every pass (8.5 us per edge)
        keep_some = use_volume and steps > 1 and not keep_vols
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
        budget_rows = int(budget / per_edge)  # pyramids (one per edge and view) the budget holds
        max_edges = max(8, min(int(os.environ.get("VIPE_AMD_BACKEND_CHUNK_EDGES", "4096")), budget_rows // V))
        if keep_some:  # finer chunks: one eighth of the budget each, so that seven of eight budget shares can stay resident
            max_edges = max(8, min(max_edges, max(256, budget_rows // (8 * V))))
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
        max_rows = max(int(sum(cnt[g0:g0 + 8].sum() for g0 in grp)) for grp in groups) * V  # the largest chunk's pyramids
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

        vols, kept_rows = {}, 0
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
                        corr_n = None  # (holds the previous chunk's levels) they go back to the allocator BEFORE this chunk's are taken
                        vol = CorrBlock.from_buffer(buf.flattened_fmaps, c["pis"] * V + c["qis"], c["pjs"] * V + c["qjs"],
                                                    frame_range=(0, t * V) if (self.cross_view and V > 1) else
                                                    (int(min(ii_np.min(), jj_np.min())) * V,
                                                     (int(max(ii_np.max(), jj_np.max())) + 1) * V))
                        WORK["pyramids_built"] += n
                        if keep_vols or (keep_some and kept_rows + n + max_rows <= budget_rows):
                            vols[gi] = vol
                            kept_rows += n
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
