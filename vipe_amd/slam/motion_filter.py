"""Keyframe selection on every incoming frame - host-side mirror of `MotionFilter`
(vipe/slam/components/motion_filter.py:27-150) and of the `DroidNet` container it drives
(vipe/slam/networks/droid_net.py:503-552).  SURVEY 8(f) row 2.

Per frame: feature encoder (HIP, `encoders.py`) -> correlation pyramid between the last keyframe's features and the
new ones -> ONE application of the flow-update operator on the identity grid -> mean flow magnitude against the
threshold; the context encoder runs only when the frame becomes a keyframe.  Everything that is constant while the
last keyframe stays (its hidden state / context features in channels-last form, the hoisted gate-context term of the
GRU) is prepared once per keyframe.  The sparse-track score (`motion_filter.py:112-135`) is host arithmetic on what a caller-supplied
tracker reports (`get_correspondences`, `get_observations`); the trackers themselves are outside the path."""
import torch

from .._lib import check, lib, ptr, require, stream_ptr
from .encoders import DroidEncoders, normalize_images
from .networks import CorrBlock, UpdateModule


class DroidNet(DroidEncoders):
    """fnet + cnet + update (droid_net.py:503-509).  The reference constructor downloads `droid.pth`; here weights
    are random-initialised unless `load_weights(path)` is given a local checkpoint (there is no network)."""

    def __init__(self):
        super().__init__()
        self.update = UpdateModule()
        self.eval()

    def load_weights(self, ckpt_path):
        """droid_net.py:529-552: strip `module.`, keep the first 2 output channels of the flow / weight heads.
        `weights_only=True`: nothing in the file is executed."""
        sd = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        sd = {k.replace("module.", ""): v for k, v in sd.items()}
        for k in ("update.weight.2.weight", "update.weight.2.bias", "update.delta.2.weight", "update.delta.2.bias"):
            sd[k] = sd[k][:2]
        # norm layers of the reference hold no parameters for norm_fn instance / none; ignore running-stat leftovers
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        require(not missing, f"checkpoint lacks {missing[:4]}...")
        self.load_state_dict({k: sd[k] for k in own})
        self.update._engine = None
        self.eval()
        return self


class MotionFilter:
    def __init__(self, droid_net, sparse_tracks=None, thresh=2.5, device=torch.device("cuda")):
        self.net = droid_net
        self.thresh = thresh
        self.device = device
        self.sparse_tracks = sparse_tracks
        self.initialized = False
        self.last_score = None
        self._pending = None  # handle of a prefetched `begin` (see prefetch)
        self.scores = []      # the dense score of every frame checked after the first (host floats; diagnostics)

    @staticmethod
    def coords_grid(ht, wd, **kwargs):
        y, x = torch.meshgrid(torch.arange(ht).to(**kwargs).float(), torch.arange(wd).to(**kwargs).float(), indexing="ij")
        return torch.stack([x, y], dim=-1)

    def _set_keyframe(self, images, x4, gmap, buffer_masks):
        net, inp = self.net.encode_context(images, x4)
        self.f_net, self.f_inp, self.f_fmap = net, inp, gmap
        self.f_mask = buffer_masks
        V, _, ht, wd = net.shape
        eng = self._engine(net.device)
        # channels-last state of the keyframe side of the one-iteration flow estimate, prepared once per keyframe
        self._net_nhwc = net.permute(0, 2, 3, 1).contiguous()
        self._xbuf = torch.empty((V, ht, wd, 320), dtype=torch.float16, device=net.device)
        self._xbuf[..., :128] = inp.permute(0, 2, 3, 1)
        self._pgate = eng.gate_context(self._xbuf) if eng.supports_gate_split(ht, wd) else None
        m0 = getattr(self, "_motn0", None)  # zero motion features (read only): built once per map size
        if m0 is None or tuple(m0.shape) != (V, ht, wd, 4) or m0.device != net.device:
            self._motn0 = torch.zeros((V, ht, wd, 4), dtype=torch.float16, device=net.device)
        c0 = getattr(self, "_coords0", None)  # the identity grid only depends on the map size: built once
        if c0 is None or tuple(c0.shape) != (V, ht, wd, 2) or c0.device != net.device:
            self._coords0 = self.coords_grid(ht, wd, device=net.device)[None].repeat(V, 1, 1, 1).contiguous()

    @torch.no_grad()
    def check(self, images, buffer_masks=None):
        """images [V,3,H,W] fp32 RGB in [0,1] on the device; buffer_masks [V,h,w] bool (True = invalid) or None.
        Returns True when the frame is to become a keyframe (its features are then in f_fmap / f_net / f_inp)."""
        h, self._pending = getattr(self, "_pending", None), None
        if h is None or h["images"] is not images:  # nothing (or another frame) was prefetched: the whole check now
            return self.finish(self.begin(images, buffer_masks))
        kept = self.finish(h)
        side = h["stream"]
        if kept and side is not None:
            # the keyframe's features / context were produced on the filter's stream: the caller's stream reads them next
            main = torch.cuda.current_stream(self.f_fmap.device)
            main.wait_stream(side)
            for x in (self.f_fmap, self.f_net, self.f_inp):
                x.record_stream(main)
        return kept

    @torch.no_grad()
    def prefetch(self, images, buffer_masks=None, stream=None):
        """Two-stage pipeline of a streaming system: ENQUEUE the first half of `check(images, buffer_masks)` now - on
        `stream`, a side stream that first joins the current one - and let the next `check` of these very tensors only
        collect the score.  The filter of frame f+1 depends on nothing but the last keyframe's features, which `check(f)`
        has already installed, so a caller that prefetches frame f+1 BEFORE it optimises keyframe f (`SLAMFrontend.run`)
        has the filter fill the chip the frontend's single-workgroup solves and small grids leave idle.  A `check` of
        other tensors drops the prefetched work and starts over (same result, no overlap)."""
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream(images.device))
            images.record_stream(stream)
            if buffer_masks is not None:
                buffer_masks.record_stream(stream)
        self._pending = self.begin(images, buffer_masks, stream=stream)

    @torch.no_grad()
    def begin(self, images, buffer_masks=None, stream=None):
        """First half of `check`: feature encoder + one flow-update application against the last keyframe, the score on
        its way to pinned host memory - everything is ENQUEUED (on `stream`, default the current one), nothing is
        waited for.  A pipeline that filters frame t+1 on a side stream while the frontend optimises keyframe t on the
        main stream calls begin(t+1) before `frontend.run()` and finish(...) after it: the filter only depends on the
        last keyframe's features, which `finish(t)` has already installed."""
        ctx = torch.cuda.stream(stream) if stream is not None else _null_ctx()
        with ctx:
            x4 = normalize_images(images)
            gmap = self.net.encode_features(images, x4)
            h = dict(images=images, x4=x4, gmap=gmap, masks=buffer_masks, stream=stream, score=None, event=None)
            if self.initialized:
                eng = self._engine(gmap.device)
                corr = CorrBlock(self.f_fmap[None], gmap[None]).lookup_deferred(self._coords0)
                _, dw, _, _ = eng.forward_nhwc(self._net_nhwc, self._xbuf, corr, self._motn0, pgate=self._pgate)
                # mean |flow| per view over the usable pixels (fp16 head output, norm in fp32: autocast rules) in one
                # launch; the minimum over the views is taken on the host once the scores have arrived
                V, ht, wd, _ = dw.shape
                score = torch.empty(V, dtype=torch.float32, device=dw.device)
                mask = self.f_mask.contiguous() if self.f_mask is not None else None
                check(lib().vipe_flow_score(ptr(dw), ptr(mask), ptr(score), V, ht * wd, stream_ptr(dw)), "flow_score")
                host = torch.empty(V, dtype=torch.float32, pin_memory=True)
                host.copy_(score, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                h["score"], h["event"] = host, ev
        return h

    @torch.no_grad()
    def finish(self, h):
        """Second half of `check`: wait for the score (the filter's stream only), decide, and on a keyframe run the
        context encoder and install the frame as the new reference (on the same stream as `begin`)."""
        ctx = torch.cuda.stream(h["stream"]) if h["stream"] is not None else _null_ctx()
        with ctx:
            if not self.initialized:
                self._set_keyframe(h["images"], h["x4"], h["gmap"], h["masks"])
                self.current_frame_idx = 0
                self.last_kf_frame_idx = 0
                self.last_n_sparse_tracks = 0
                self.initialized = True
                return True
            self.current_frame_idx += 1
            h["event"].synchronize()
            self.last_score = float(h["score"].min())
            self.scores.append(self.last_score)
            sparse = self._sparse_motion_score(h["images"].shape[0])
            # the track score is a sum over keypoints' mean displacement, not a pixel average: twice the threshold
            if self.last_score > self.thresh or sparse > self.thresh * 2:
                self._set_keyframe(h["images"], h["x4"], h["gmap"], h["masks"])
                self.last_kf_frame_idx = self.current_frame_idx
                self.last_n_sparse_tracks = 0
                return True
            return False

    def _sparse_motion_score(self, n_views):
        """motion_filter.py:112-135, host arithmetic on what the caller's tracker reports: mean displacement of the
        keypoints seen in both the current frame and the last keyframe, summed over the views; + 100 when more than 20 %
        of the tracks of the previous frame are gone (forces a keyframe)."""
        st = self.sparse_tracks
        if st is None or not getattr(st, "enabled", False):
            return 0.0
        score, n_tracks = 0.0, 0
        for v in range(n_views):
            kp = st.get_correspondences(v, self.current_frame_idx, self.last_kf_frame_idx)
            n_tracks += len(kp)
            cur = st.get_observations(v, self.current_frame_idx, kp)
            last = st.get_observations(v, self.last_kf_frame_idx, kp)
            score += (cur - last).norm(dim=-1).mean().item()  # no common keypoint: nan, which never exceeds a threshold
        lost = self.last_n_sparse_tracks - n_tracks
        if lost > 0 and self.last_n_sparse_tracks > 0 and lost / self.last_n_sparse_tracks > 0.2:
            score += 100.0
        self.last_n_sparse_tracks = n_tracks
        self.last_sparse_score = score
        return score

    def _engine(self, device):
        """The filter's own execution engine of the update operator: private scratch buffers and descriptors, so that
        a filter running on a side stream never shares them with the factor graph's iterations on the main stream."""
        if getattr(self, "_eng", None) is None or self._eng.device != device:
            from .update_engine import UpdateEngine
            self._eng = UpdateEngine(self.net.update, device)
        return self._eng


class _null_ctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
