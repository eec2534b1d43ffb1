"""Clip-level data parallelism: the reference runs clips in a sequential Python loop (run.py:17-26);
here clip i goes to rank i mod world, every rank owns one GPU and its own SLAM state, and the only exchange is ONE
all_gather of fixed-shape padded results (trajectory [F_max,7] fp32 + valid length + intrinsics [4] + status) at the
end - ~10 KB per clip, latency-bound, so RCCL over xGMI is used for nothing else.  A failed clip is reported, the
others continue (per-clip isolation).
"""

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class ClipResult:
    clip_id: int
    poses: torch.Tensor  # [F,7] world->camera (tx,ty,tz,qx,qy,qz,qw)
    intrinsics: torch.Tensor  # [4]
    ok: bool = True
    seconds: float = 0.0


def shard_clips(n_clips, rank, world):
    """Round-robin assignment: balanced to within one clip, independent of clip order on disk."""
    return list(range(rank, n_clips, world))


def _pack(results, n_slots, f_max, device):
    """-> float32 [n_slots, 7 + f_max*7]: (clip_id, ok, n_frames, seconds, fx,fy,cx... ) header then the trajectory."""
    buf = torch.zeros((n_slots, 8 + f_max * 7), dtype=torch.float32, device=device)
    buf[:, 0] = -1.0
    for s, r in enumerate(results):
        f = min(int(r.poses.shape[0]), f_max)
        buf[s, 0] = float(r.clip_id)
        buf[s, 1] = 1.0 if r.ok else 0.0
        buf[s, 2] = float(f)
        buf[s, 3] = float(r.seconds)
        buf[s, 4:8] = r.intrinsics.to(device=device, dtype=torch.float32)
        buf[s, 8:8 + 7 * f] = r.poses[:f].to(device=device, dtype=torch.float32).reshape(-1)
    return buf


def _unpack(buf, f_max):
    out = []
    for row in buf:
        cid = int(row[0].item())
        if cid < 0:
            continue
        f = int(row[2].item())
        out.append(ClipResult(cid, row[8:8 + 7 * f].reshape(f, 7).cpu(), row[4:8].cpu(), bool(row[1].item() > 0.5),
                              float(row[3].item())))
    return sorted(out, key=lambda r: r.clip_id)


def gather_results(local_results, n_clips, f_max, device=None):
    """All ranks call this once; every rank receives the full, clip-ordered result list."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    device = device or (local_results[0].poses.device if local_results else torch.device("cpu"))
    n_slots = (n_clips + world - 1) // world
    mine = _pack(local_results, n_slots, f_max, device)
    if world == 1:
        return _unpack(mine, f_max)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return _unpack(torch.cat(parts, 0), f_max)


def run_sharded(n_clips, process_clip, f_max, device=None):
    """process_clip(clip_id) -> ClipResult.  Exceptions are confined to their clip."""
    import time

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
    return gather_results(results, n_clips, f_max, device)
