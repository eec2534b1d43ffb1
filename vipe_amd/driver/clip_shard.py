"""Clip-level data parallelism: the reference runs clips in a sequential Python loop (run.py:17-26);
here clip i goes to rank i mod world, every rank owns one GPU and its own SLAM state, and the only exchange is ONE
all_gather of fixed-shape padded results (trajectory [F_max,7] fp32 + valid length + intrinsics [4] + status) at the
end - ~10 KB per clip, latency-bound, so RCCL over xGMI is used for nothing else.  A failed clip is reported, the
others continue (per-clip isolation).
"""

import time
from dataclasses import dataclass

import torch
import torch.distributed as dist

HEADER = 8  # clip_id, ok, n_frames (valid rows), seconds, fx, fy, cx, cy


@dataclass
class ClipResult:
    clip_id: int
    poses: torch.Tensor  # [F,7] world->camera (tx,ty,tz,qx,qy,qz,qw)
    intrinsics: torch.Tensor  # [4]
    ok: bool = True
    seconds: float = 0.0
    truncated: bool = False  # the trajectory had more than f_max rows: only the first f_max travelled


def shard_clips(n_clips, rank, world):
    """Round-robin assignment: balanced to within one clip, independent of clip order on disk."""
    return list(range(rank, n_clips, world))


def exchange_device():
    """The device the collective's buffers must live on.  It is a property of the process group, never of the results
    (a rank whose shard is empty, or whose first clip failed, holds no device tensor at all): RCCL ("nccl") moves
    device memory of the rank's current GPU, gloo moves host memory."""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _pack(results, n_slots, f_max, device):
    """-> float32 [n_slots, 8 + f_max*7]: header (clip_id, ok, n_frames, seconds, intrinsics; n_frames < 0 flags a
    truncated trajectory) then the trajectory.  Assembled on the host, moved to `device` with one copy."""
    buf = torch.zeros((n_slots, HEADER + f_max * 7), dtype=torch.float32)
    buf[:, 0] = -1.0
    for s, r in enumerate(results):
        F = int(r.poses.shape[0])
        f = min(F, f_max)
        buf[s, 0] = float(r.clip_id)
        buf[s, 1] = 1.0 if r.ok else 0.0
        buf[s, 2] = float(-f if F > f_max else f)
        buf[s, 3] = float(r.seconds)
        buf[s, 4:8] = r.intrinsics.detach().to(device="cpu", dtype=torch.float32).reshape(-1)[:4]
        buf[s, HEADER:HEADER + 7 * f] = r.poses[:f].detach().to(device="cpu", dtype=torch.float32).reshape(-1)
    return buf.to(device)


def _unpack(buf):
    out = []
    for row in buf.cpu():  # one device-to-host copy, then host parsing
        cid = int(row[0])
        if cid < 0:
            continue
        f = int(row[2])
        out.append(ClipResult(cid, row[HEADER:HEADER + 7 * abs(f)].reshape(abs(f), 7).clone(), row[4:8].clone(),
                              bool(row[1] > 0.5), float(row[3]), truncated=f < 0))
    return sorted(out, key=lambda r: r.clip_id)


def gather_results(local_results, n_clips, f_max, device=None, strict=True):
    """All ranks call this once; every rank receives the full, clip-ordered result list.  A trajectory longer than
    f_max travels truncated with a flag in its header; with `strict` every rank raises on it AFTER the exchange (all
    ranks see the flag, so none is left waiting in the collective)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    device = device or exchange_device()
    n_slots = (n_clips + world - 1) // world
    assert len(local_results) <= n_slots, "more local results than this rank's share of the clips"
    mine = _pack(local_results, n_slots, f_max, device)
    if world == 1:
        res = _unpack(mine)
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        res = _unpack(torch.cat(parts, 0))
    if strict and any(r.truncated for r in res):
        raise ValueError(f"trajectories of clips {[r.clip_id for r in res if r.truncated]} exceed f_max={f_max} rows")
    return res


def run_sharded(n_clips, process_clip, f_max, device=None, strict=True):
    """process_clip(clip_id) -> ClipResult.  Exceptions are confined to their clip: the failed clip travels as an
    empty trajectory with ok = False and the rank still enters the collective with everyone else."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    results = []
    for cid in shard_clips(n_clips, rank, world):
        t0 = time.perf_counter()
        try:
            r = process_clip(cid)
            r.seconds = time.perf_counter() - t0
        except Exception:  # noqa: BLE001 - isolate the clip, report it
            r = ClipResult(cid, torch.zeros(0, 7), torch.zeros(4), ok=False, seconds=time.perf_counter() - t0)
        results.append(r)
    return gather_results(results, n_clips, f_max, device, strict)
