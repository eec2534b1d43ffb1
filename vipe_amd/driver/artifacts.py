"""Result artifacts in the reference's on-disk formats (vipe/utils/io.py:144-225), so that clips processed here can be
compared with / consumed like the reference's outputs.  SURVEY 8(f) row 4.

  pose/<name>.npz        data [F,4,4] float32 OpenCV cam2world matrices, inds [F] frame indices (io.py:144-162)
  intrinsics/<name>.npz  data [F,4|5] float32 [fx,fy,cx,cy(,k1)], inds [F]                       (io.py:181-203)
  intrinsics/<name>_camera.txt   "<frame_idx>: <CAMERA_TYPE>" per line                            (io.py:205-214)
Depth maps (zipped half-float EXR, io.py:250-277) need OpenEXR, which this image does not have."""
import os

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
