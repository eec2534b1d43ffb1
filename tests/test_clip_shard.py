"""N>1 path on CPU: world_size-2 gloo processes, clip sharding + the single result gather."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vipe_amd.driver.clip_shard import ClipResult, gather_results, run_sharded, shard_clips


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def process(cid):
        if cid == 3:
            raise ValueError("malformed clip")  # per-clip isolation (system.py:298-299 analogue)
        f = 5 + cid  # ragged trajectory lengths
        poses = torch.zeros(f, 7)
        poses[:, 0] = cid
        poses[:, 1] = torch.arange(f)
        poses[:, 6] = 1
        return ClipResult(cid, poses, torch.tensor([100.0 + cid, 100.0, 64.0, 48.0]))

    res = run_sharded(n_clips, process, f_max=16)
    q.put((rank, [(r.clip_id, r.ok, tuple(r.poses.shape), float(r.poses[:, 1].sum()) if r.ok else -1.0,
                   float(r.intrinsics[0])) for r in res]))
    dist.barrier()
    dist.destroy_process_group()


def _worker_edge(rank, world, port, q):
    """3 clips on 4 ranks: rank 3's shard is empty; clip 0 (rank 0's FIRST clip) fails; clip 2 is longer than f_max."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def process(cid):
        if cid == 0:
            raise RuntimeError("decoder failed")
        poses = torch.zeros(4 if cid == 1 else 9, 7)
        poses[:, 6] = 1
        return ClipResult(cid, poses, torch.tensor([1.0, 2.0, 3.0, 4.0]))

    res = run_sharded(3, process, f_max=6, strict=False)
    strict_raises = False
    try:
        run_sharded(3, process, f_max=6, strict=True)
    except ValueError:
        strict_raises = True
    q.put((rank, [(r.clip_id, r.ok, tuple(r.poses.shape), r.truncated) for r in res], strict_raises))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_with_empty_shard_failed_first_clip_and_overlong_trajectory():
    """the exchange device comes from the process group, not from the results, so a rank that holds no
    tensor at all (empty shard / failed first clip) still enters the collective; F > f_max is flagged, never silent."""
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_edge, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, got, strict_raises in outs:
        assert got == [(0, False, (0, 7), False), (1, True, (4, 7), False), (2, True, (6, 7), True)]
        assert strict_raises, "strict mode must raise on every rank after the exchange"


def test_exchange_device_without_process_group_is_cpu():
    from vipe_amd.driver.clip_shard import exchange_device
    assert exchange_device() == torch.device("cpu")


def test_shard_assignment_is_a_partition():
    for n, w in [(8, 8), (5, 2), (1, 4), (0, 3), (17, 8)]:
        parts = [shard_clips(n, r, w) for r in range(w)]
        flat = sorted(c for p in parts for c in p)
        assert flat == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_two_rank_gloo_gather():
    world, n_clips = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[0] == outs[1], "every rank must receive the same gathered list"
    got = outs[0]
    assert [g[0] for g in got] == [0, 1, 2, 3, 4]
    for cid, ok, shape, s, fx in got:
        if cid == 3:
            assert not ok and shape == (0, 7)
        else:
            f = 5 + cid
            assert ok and shape == (f, 7) and s == sum(range(f)) and fx == 100.0 + cid


def test_single_process_gather_is_identity():
    r = [ClipResult(1, torch.ones(3, 7), torch.arange(4.0)), ClipResult(0, torch.zeros(2, 7), torch.zeros(4))]
    out = gather_results(r, n_clips=2, f_max=4)
    assert [o.clip_id for o in out] == [0, 1] and out[1].poses.shape == (3, 7)


def test_pose_and_intrinsics_artifacts_round_trip(tmp_path):
    """Reference on-disk formats (vipe/utils/io.py:144-225): `data` = OpenCV cam2world 4x4 float32, `inds` = frame ids."""
    import numpy as np
    from oracle import se3 as ose3
    from vipe_amd.driver import artifacts
    from vipe_amd.driver.clip_shard import ClipResult
    rng = np.random.default_rng(0)
    w2c = ose3.se3_exp(0.3 * rng.standard_normal((5, 6)))
    p = tmp_path / "pose" / "a.npz"
    artifacts.save_pose_artifacts(str(p), w2c, inds=[0, 2, 4, 6, 8])
    raw = np.load(p)
    assert sorted(raw.files) == ["data", "inds"] and raw["data"].dtype == np.float32 and raw["data"].shape == (5, 4, 4)
    inds, c2w = artifacts.read_pose_artifacts(str(p))
    assert list(inds) == [0, 2, 4, 6, 8]
    ref = np.linalg.inv(ose3.se3_matrix(w2c))
    assert np.allclose(c2w, ref, atol=1e-5)
    ip = tmp_path / "intrinsics" / "a.npz"
    cp = tmp_path / "intrinsics" / "a_camera.txt"
    artifacts.save_intrinsics_artifacts(str(ip), np.tile([460.8, 460.8, 256.0, 192.0], (5, 1)), camera_path=str(cp))
    inds, K, types = artifacts.read_intrinsics_artifacts(str(ip), str(cp))
    assert K.shape == (5, 4) and K.dtype == np.float32 and types == ["PINHOLE"] * 5 and list(inds) == list(range(5))
    assert cp.read_text().splitlines()[3] == "3: PINHOLE"
    res = [ClipResult(3, torch.from_numpy(w2c).float(), torch.tensor([1.0, 2.0, 3.0, 4.0])),
           ClipResult(4, torch.zeros(0, 7), torch.zeros(4), ok=False)]
    assert artifacts.save_clip_results(str(tmp_path / "out"), res) == ["clip_00003"]
    assert np.allclose(artifacts.read_pose_artifacts(str(tmp_path / "out" / "pose" / "clip_00003.npz"))[1], ref, atol=1e-5)


def test_corr_pool_slot_bookkeeping_on_host():
    """CorrPool (vipe_amd/slam/networks.py): slots are handed out, released and reused without moving the level buffers;
    `corr_pyramid` materialises the edge-ordered levels.  Pure tensor bookkeeping - runs on CPU tensors."""
    import types
    import numpy as np
    from vipe_amd.slam.networks import CorrPool

    def block(n, tag):
        lv = [torch.full((n, 2, 2, 4 >> i, 4 >> i), float(tag), dtype=torch.float16) + torch.arange(n).view(n, 1, 1, 1, 1)
              for i in range(3)]
        return types.SimpleNamespace(corr_pyramid=lv)

    pool = CorrPool(num_levels=3, capacity=4)
    pool.cat(block(3, 10))                      # edges 0,1,2 -> slots 0,1,2
    assert pool._slots_host == [0, 1, 2] and pool._free == [3] and len(pool) == 3
    pool = pool[np.array([0, 2])]               # drop edge 1 -> slot 1 is free again
    assert pool._slots_host == [0, 2] and sorted(pool._free) == [1, 3]
    pool.cat(block(2, 20))                      # reuses slots 1 and 3
    assert pool._slots_host[:2] == [0, 2] and sorted(pool._slots_host[2:]) == [1, 3] and pool._free == []
    pool.cat(block(3, 30))                      # grows: capacity doubles until 3 more fit
    assert pool.pool[0].shape[0] == 8 and pool._slots_host[-3:] == [4, 5, 6] and pool._free == [7]
    want = [10, 12, 20, 21, 30, 31, 32]
    for lv in pool.corr_pyramid:
        assert lv.shape[0] == 7 and [float(x) for x in lv[:, 0, 0, 0, 0]] == want
    assert pool.slots.dtype == torch.int32 and pool.slots.tolist() == pool._slots_host
    pool = pool[torch.tensor([True, False, True, False, True, False, True])]  # boolean masks work too
    assert [float(x) for x in pool.corr_pyramid[0][:, 0, 0, 0, 0]] == [10, 20, 30, 32]
    assert len(pool._free) == 4 and set(pool._free).isdisjoint(pool._slots_host) and len(set(pool._slots_host)) == 4


def test_corr_pool_first_block_larger_than_default_capacity():
    """A first block larger than the pool's default capacity sizes the pool to fit; later blocks grow it.  (The
    product path, `add_edges`, has the build kernel write into the pool's slots directly - no copy to avoid.)"""
    import types
    import numpy as np
    from vipe_amd.slam.networks import CorrPool

    def block(n, tag):
        lv = [torch.full((n, 2, 2, 4 >> i, 4 >> i), float(tag), dtype=torch.float16) + torch.arange(n).view(n, 1, 1, 1, 1)
              for i in range(2)]
        return types.SimpleNamespace(corr_pyramid=lv)

    pool = CorrPool(num_levels=2, capacity=4).cat(block(5, 10))
    assert pool.pool[0].shape[0] == 8 and pool._slots_host == [0, 1, 2, 3, 4] and pool._free == [5, 6, 7]
    pool.cat(block(4, 20))  # one slot short: the pool grows
    assert pool.pool[0].shape[0] == 16 and len(pool) == 9
    assert [float(x) for x in pool.corr_pyramid[0][:, 0, 0, 0, 0]] == [10, 11, 12, 13, 14, 20, 21, 22, 23]
    pool = pool[np.array([1, 5])]
    assert [float(x) for x in pool.corr_pyramid[1][:, 0, 0, 0, 0]] == [11, 20] and len(pool._free) == pool.pool[0].shape[0] - 2


def test_blocked_pyramid_layout_round_trip():
    """VIPE_PYRAMID_BLOCKED <-> reference layout converters are inverse permutations, and element (p1, y, x) of the
    reference tensor sits where include/vipe_amd.h says it does in the blocked one."""
    from vipe_amd.ext import droid_net_ext as dn
    from vipe_amd.slam.networks import _to_blocked
    h, w, n = 8, 64, 2
    ref = [torch.arange(n * h * w * (h >> i) * (w >> i), dtype=torch.float32).reshape(n, h, w, h >> i, w >> i) for i in range(4)]
    blk = _to_blocked(ref, h, w)
    assert [tuple(b.shape) for b in blk] == dn.pyramid_level_shapes(n, h, w, 4, dn.BLOCKED)
    back = dn.pyramid_to_reference(blk, h, w)
    assert all(torch.equal(a, b) for a, b in zip(ref, back))
    for (e, p1, y, x) in [(0, 0, 0, 0), (1, 77, 5, 43), (1, 511, 7, 63), (0, 130, 3, 8)]:
        assert blk[0][e, p1 // 64, (x // 32) * (h // 4) + y // 4, p1 % 64, (x % 32) // 8, y % 4, x % 8] == \
            ref[0][e, p1 // w, p1 % w, y, x]
        y1, x1 = y // 2, x // 2
        assert blk[1][e, p1 // 64, (x1 // 16) * (h // 8) + y1 // 4, p1 % 64, (x1 % 16) // 8, y1 % 4, x1 % 8] == \
            ref[1][e, p1 // w, p1 % w, y1, x1]

def test_bench_gpus_flag_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (no external launcher) must start two rank processes, run the
    result gather and print ONE line with n_gpus = 2.  `--mode plumbing` is the launch / process-group / gather /
    artifact control flow of the video mode with fake clips, so it runs without a GPU (gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "plumbing", "--clips", "5",
                        "--frames", "12", "--out-dir", str(tmp_path / "art")], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["clip_ids"] == [0, 1, 2, 3, 4] and out["config"]["artifacts_written"] == 5
    assert sorted(os.listdir(tmp_path / "art" / "pose")) == [f"clip_{i:05d}.npz" for i in range(5)]
    # a launcher / flag mismatch is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "plumbing"],
                       capture_output=True, text=True, env=env2, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_depth_artifacts_round_trip_and_exr_layout(tmp_path):
    """Depth artifacts (vipe/utils/io.py:250-310): zipped single-channel half-float EXR files.  No OpenEXR here, so the
    codec is checked by round trips (both chunk codings, ragged last chunk), by the header bytes the file-layout
    document prescribes, and by the reader's failure mode (an unreadable member -> NaN map of the last good size)."""
    import struct
    import zipfile
    import numpy as np
    from vipe_amd.driver import artifacts, exr
    rng = np.random.default_rng(0)
    d = (1.0 + 9.0 * rng.random((3, 37, 53))).astype(np.float32)
    d[1, 5:9] = 65504.0  # largest half
    for comp in (exr.NO_COMPRESSION, exr.ZIP_COMPRESSION, exr.ZIPS_COMPRESSION):
        blob = exr.write_exr_half(d[0], comp)
        assert struct.unpack_from("<ii", blob, 0) == (20000630, 2)
        assert blob[8:8 + 9] == b"channels\0" and b"compression\0compression\0" in blob and b"dataWindow\0box2i\0" in blob
        back = exr.read_exr_half(blob)
        assert back.dtype == np.float16 and np.array_equal(back, d[0].astype(np.float16))
    assert len(exr.write_exr_half(np.ones((64, 64)), exr.ZIP_COMPRESSION)) < 64 * 64 * 2 // 4  # deflate does its job
    p = tmp_path / "depth" / "clip.zip"
    artifacts.save_depth_artifacts(str(p), [torch.from_numpy(d[0]), None, torch.from_numpy(d[2])])
    with zipfile.ZipFile(p) as z:
        assert sorted(z.namelist()) == ["00000.exr", "00002.exr"] and z.infolist()[0].compress_type == zipfile.ZIP_DEFLATED
    got = list(artifacts.read_depth_artifacts(str(p)))
    assert [g[0] for g in got] == [0, 2] and got[0][1].dtype == torch.float32
    assert torch.equal(got[1][1], torch.from_numpy(d[2].astype(np.float16).astype(np.float32)))
    with zipfile.ZipFile(p, "a") as z:
        z.writestr("00003.exr", b"not an exr file")
    got = list(artifacts.read_depth_artifacts(str(p)))
    assert got[2][0] == 3 and got[2][1].shape == (37, 53) and bool(torch.isnan(got[2][1]).all())
    # truncated / corrupt members (cut inside the header, the offset table, a compressed chunk) are unreadable frames too,
    # not a crash of the whole iterator
    good = exr.write_exr_half(d[2], exr.ZIP_COMPRESSION)
    for cut in (6, 40, len(good) // 2, len(good) - 3):
        with pytest.raises(OSError):
            exr.read_exr_half(good[:cut])
    flat = bytearray(exr.write_exr_half(np.ones((64, 64)), exr.ZIP_COMPRESSION))  # compressible: chunks are deflate streams
    flat[-12:-4] = b"\xff" * 8  # garbage inside the last deflate stream
    with pytest.raises(OSError):
        exr.read_exr_half(bytes(flat))
    with zipfile.ZipFile(p, "a") as z:
        z.writestr("00004.exr", good[:len(good) // 2])
    got = list(artifacts.read_depth_artifacts(str(p)))
    assert [g_[0] for g_ in got] == [0, 2, 3, 4] and bool(torch.isnan(got[3][1]).all())


def test_factor_graph_index_tensors_follow_the_host_mirror():
    """`FactorGraph.ii / jj / age / ii_inac / jj_inac` are device views of the host mirror made on demand
    (factor_graph._index_property): edits of the mirror invalidate them, an assignment from outside makes the mirror
    follow the tensor."""
    import numpy as np
    import torch
    from vipe_amd.slam.factor_graph import FactorGraph

    g = object.__new__(FactorGraph)
    g.device = torch.device("cpu")
    g.ii, g.jj = torch.tensor([3, 4, 5]), torch.tensor([1, 2, 3])
    g.ii_inac = g.jj_inac = torch.zeros(0, dtype=torch.long)
    h = g.host_edges()
    assert h["ii"].tolist() == [3, 4, 5] and h["age"].tolist() == [0, 0, 0] and h["ii_inac"].shape == (0,)
    h["age"] += 2
    h["ii"] = np.array([7, 8], dtype=np.int64)
    h["jj"] = np.array([5, 6], dtype=np.int64)
    h["age"] = h["age"][:2]
    g._mirror_changed("ii", "jj", "age")
    assert g.ii.tolist() == [7, 8] and g.jj.tolist() == [5, 6] and g.age.tolist() == [2, 2]
    assert g.ii is g.ii  # cached until the mirror changes again
    g.ii = torch.tensor([1])  # replaced from outside: the mirror is rebuilt from the tensors
    g.jj = torch.tensor([0])
    assert g.host_edges()["ii"].tolist() == [1] and g.host_edges()["age"].tolist() == [0]
