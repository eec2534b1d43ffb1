"""Result artifacts in the reference's on-disk formats (vipe/utils/io.py:144-225), so that clips processed here can be
compared with / consumed like the reference's outputs.  SURVEY 8(f) row 4.

  pose/<name>.npz        data [F,4,4] float32 OpenCV cam2world matrices, inds [F] frame indices (io.py:144-162)
  intrinsics/<name>.npz  data [F,4|5] float32 [fx,fy,cx,cy(,k1)], inds [F]                       (io.py:181-203)
  intrinsics/<name>_camera.txt   "<frame_idx>: <CAMERA_TYPE>" per line                            (io.py:205-214)
  depth/<name>.zip       one `<frame_idx:05d>.exr` per frame: a single HALF channel `Z` (metric depth), ZIP_DEFLATED
                         archive (io.py:250-277).  The OpenEXR bindings are absent here: the container is written and
                         read by `vipe_amd/driver/exr.py` from the published file layout (parity unpinned)."""
import os
import zipfile

import numpy as np
import torch

from ..ext.lietorch import SE3


def cam2world_matrices(poses_w2c):
    """[F,7] world->camera rows (the SLAM buffer's convention) -> [F,4,4] float32 cam2world"""
    return SE3(torch.as_tensor(poses_w2c, dtype=torch.float32)).inv().matrix().cpu().numpy().astype(np.float32)


def save_pose_artifacts(path, poses_w2c, inds=None):
    data = cam2world_matrices(poses_w2c)
    inds = np.arange(data.shape[0]) if inds is None else np.asarray(inds)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, data=data, inds=inds)


def read_pose_artifacts(path):
    """-> (inds [F], cam2world [F,4,4])"""
    d = np.load(path)
    return d["inds"], d["data"]


def save_intrinsics_artifacts(path, intrinsics, inds=None, camera_type="PINHOLE", camera_path=None):
    data = np.asarray(torch.as_tensor(intrinsics).cpu().numpy(), dtype=np.float32)
    if data.ndim == 1:
        data = data[None]
    inds = np.arange(data.shape[0]) if inds is None else np.asarray(inds)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, data=data, inds=inds)
    if camera_path is not None:
        with open(camera_path, "w") as f:
            for i in inds:
                f.write(f"{int(i)}: {camera_type}\n")


def read_intrinsics_artifacts(path, camera_path=None):
    d = np.load(path)
    inds, data = d["inds"], d["data"]
    if camera_path is None or not os.path.exists(camera_path):
        assert data.shape[1] == 4
        types = ["PINHOLE"] * data.shape[0]
    else:
        types = [line.split(":")[1].strip() for line in open(camera_path)]
    return inds, data, types


def save_depth_artifacts(path, depths, inds=None):
    """io.py:250-277: metric depth maps [F,H,W] (or an iterable of [H,W], None entries skipped) -> zipped EXR files"""
    from .exr import write_exr_half
    items = [(i if inds is None else int(inds[i]), d) for i, d in enumerate(depths) if d is not None]
    if not items:
        return
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        for frame_idx, d in items:
            d = d.detach().cpu().numpy() if torch.is_tensor(d) else np.asarray(d)
            z.writestr(f"{frame_idx:05d}.exr", write_exr_half(d))


def read_depth_artifacts(path):
    """io.py:279-310 -> iterator of (frame_idx, float32 [H,W] tensor); an unreadable member yields an all-NaN map of the
    last good size, as the reference does"""
    from .exr import read_exr_half
    shape = None
    with zipfile.ZipFile(path, "r") as z:
        for name in sorted(z.namelist()):
            frame_idx = int(name.split(".")[0])
            try:
                d = read_exr_half(z.read(name))
            except (OSError, KeyError, ValueError, zipfile.BadZipFile):
                assert shape is not None
                yield frame_idx, torch.full(shape, float("nan"), dtype=torch.float32)
                continue
            shape = d.shape
            yield frame_idx, torch.from_numpy(d.astype(np.float32))


def save_clip_results(out_dir, results, name_of=lambda r: f"clip_{r.clip_id:05d}"):
    """Rank 0 after `clip_shard.gather_results`: one pose + intrinsics artifact per successful clip."""
    written = []
    for r in results:
        if not r.ok or r.poses.shape[0] == 0:
            continue
        name = name_of(r)
        save_pose_artifacts(os.path.join(out_dir, "pose", name + ".npz"), r.poses)
        save_intrinsics_artifacts(os.path.join(out_dir, "intrinsics", name + ".npz"),
                                  r.intrinsics[None].expand(r.poses.shape[0], -1),
                                  camera_path=os.path.join(out_dir, "intrinsics", name + "_camera.txt"))
        written.append(name)
    return written
