"""bench.py - dense-BA + flow update iterations per second on the BASELINE.json workload.

One "step" = one FactorGraph.update: reproject -> 4-level correlation lookup -> flow-update
operator (ConvGRU) -> dense bundle adjustment (3 Gauss-Newton iterations) on a synthetic 512x384 clip,
48 keyframes, E = 276 edges (radius-3 bidirectional graph), sensor-depth prior on (BASELINE config 3), inputs resident in
HBM.  With --gpus N every rank runs its own clip (clip sharding, no data-path collective) and the per-clip results are
exchanged with ONE all_gather (RCCL) at the end; the value is the whole-job aggregate.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1: starts its own N rank processes
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --mode video --gpus N --frames 300            # BASELINE config 4: one 300-frame clip per GPU
"""

import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0  # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0
DTYPE = "f16 (correlation, GRU; fp32 accumulate) + f32 geometry/BA (fp64 reduced system)"


_T0 = time.perf_counter()


def _log(msg):
    """progress on stderr (stdout carries the one JSON line): a silent multi-minute run is indistinguishable from a hang"""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}\n")
        sys.stderr.flush()


def host_cores():
    """CPU threads this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota (a GPU box
    hands each job a share of a many-core host; os.cpu_count() reports the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001 - cgroup v1 / not readable: the affinity mask stands
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:  # noqa: BLE001
            pass
    return n


# ---------------------------------------------------------------------------------------------- launching N ranks
def spawn_ranks(n):
    """`python bench.py --gpus N` without an external launcher: start N fresh rank processes (one per GPU) through
    torch.distributed.run and relay their output.  Runs BEFORE this process has made any GPU call (importing torch and
    counting devices do not initialise the runtime); the parent never touches the GPU, it only waits."""
    plumbing = "plumbing" in sys.argv  # --mode plumbing: gloo ranks on the CPU
    backend = "gloo" if plumbing else os.environ.get("VIPE_BENCH_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        sys.stderr.write(f"bench.py: --gpus {n} needs {n} GPUs for RCCL, this node shows {have} (set "
                         f"VIPE_BENCH_DIST_BACKEND=gloo to rehearse the multi-rank path with ranks sharing a card)\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------- the workload
def build_problem(device, n_kf, height, width, radius, extra_edges, seed=1234, depth_prior=True):
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.factor_graph import FactorGraph
    from vipe_amd.slam.networks import UpdateModule
    from vipe_amd.synth import make_graph

    g = make_graph(n=n_kf, height=height, width=width, radius=radius, extra_edges=extra_edges, seed=seed,
                   depth_prior=depth_prior)
    buf = GraphBuffer(height, width, n_views=1, buffer_size=max(64, n_kf), device=device)
    buf.n_frames = n_kf
    buf.poses[:n_kf] = torch.from_numpy(g.poses).to(device)
    buf.disps[:n_kf, 0] = torch.from_numpy(g.disps).to(device)
    buf.disps_sens[:n_kf, 0] = torch.from_numpy(g.disps_sens).to(device)  # config 3: "depth_align on"
    buf.intrinsics[:] = torch.from_numpy(g.intrinsics).to(device)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    ht, wd = g.ht, g.wd
    buf.fmaps[:n_kf, 0] = torch.randn(n_kf, 128, ht, wd, generator=gen).half().to(device)
    buf.nets[:n_kf, 0] = torch.randn(n_kf, 128, ht, wd, generator=gen).tanh().half().to(device)
    buf.inps[:n_kf, 0] = torch.randn(n_kf, 128, ht, wd, generator=gen).relu().half().to(device)
    torch.manual_seed(seed)
    um = UpdateModule().eval()
    graph = FactorGraph(um, buf, device, max_factors=-1)
    graph.add_factors(torch.from_numpy(g.ii), torch.from_numpy(g.jj))
    # start from the synthetic targets / weights so that the first BA sees the documented problem
    graph.target = torch.from_numpy(g.target).to(device)[None].contiguous()
    graph.weight = torch.from_numpy(g.weight).to(device)[None].contiguous()
    return g, buf, graph


def cpu_baseline(n_sub=8, budget_s=30.0):
    """the PyTorch-CPU corr + BA path (oracle/torch_cpu.py: correlation volume + 4-level pyramid +
    7x7 bilinear lookup, then the 3-iteration dense Schur BA; GRU excluded), timed on this box's host cores with all the
    threads the job may use (1 warm-up + median of 5) and with 1 thread.  The full N = 48 / E = 276 graph is timed when
    6 passes of it fit `budget_s` (estimated from a pass over an n_sub-keyframe sub-graph of the same clip); otherwise
    the sub-graph is the sample and the figure is scaled per edge."""
    from oracle import torch_cpu as tc
    from vipe_amd.synth import make_graph

    T = torch.from_numpy

    def problem(n):
        g = make_graph(n=n, height=384, width=512, radius=3, seed=1234, depth_prior=True)
        E = len(g.ii)
        gen = torch.Generator().manual_seed(1234)
        fm = torch.randn(n, 128, g.ht, g.wd, generator=gen).half().float()
        ii, jj = T(g.ii), T(g.jj)
        tgt = T(g.target)
        coords = tgt + 0.5 * torch.randn(tgt.shape, generator=gen)
        ba_in = (T(g.poses), T(g.disps), T(g.disps_sens), T(g.intrinsics), tgt.reshape(E, -1, 2),
                 T(g.weight).reshape(E, -1, 2), T(g.eta), ii, jj, 1, n, 3, 1e-3, 0.1)

        def one_pass():
            tb = tl = 0.0
            for c in range(0, E, 46):  # chunks of 46 edges bound the fp32 volume to ~2.3 GB
                t0 = time.perf_counter()
                pyr = tc.corr_pyramid(fm[ii[c:c + 46]], fm[jj[c:c + 46]])
                t1 = time.perf_counter()
                tc.corr_lookup(pyr, coords[c:c + 46])
                tb, tl = tb + (t1 - t0), tl + (time.perf_counter() - t1)
            t2 = time.perf_counter()
            tc.bundle_adjustment(*ba_in)
            return tb, tl, time.perf_counter() - t2

        return E, one_pass

    cores = host_cores()
    keep = torch.get_num_threads()
    torch.set_num_threads(cores)
    _log(f"cpu_baseline: {cores} threads (os.cpu_count() = {os.cpu_count()})")
    E_s, pass_s = problem(n_sub)
    pass_s()
    est_full = sum(pass_s()) * 276.0 / E_s
    full = est_full * 6.5 <= budget_s
    _log(f"cpu_baseline: estimated full-size pass {est_full:.1f}s -> {'full N=48 / E=276 graph' if full else 'sub-graph sample'}")
    E, one_pass = problem(48) if full else (E_s, pass_s)
    if full:
        one_pass()
    runs = sorted((one_pass() for _ in range(5)), key=sum)
    b, l, a = runs[len(runs) // 2]
    torch.set_num_threads(1)
    b1, l1, a1 = pass_s()
    torch.set_num_threads(keep)
    scale = 276.0 / E
    return {
        "value": 1.0 / ((b + l + a) * scale), "unit": "iters/s", "cores": cores, "kind": "port",
        "value_1_thread": 1.0 / ((b1 + l1 + a1) * 276.0 / E_s),
        "value_without_volume_build": 1.0 / ((l + a) * scale),
        "seconds_per_pass": {"edges": E, "volume+pyramid build": b, "lookup": l, "dense BA (3 GN iterations)": a},
        "sample": f"PyTorch-CPU (fp32) correlation volume + pyramid + 7x7x4 lookup + 3-iteration dense Schur BA, GRU "
                  f"excluded, on " + (f"the full 48-keyframe / {E}-edge bench graph" if full else
                                      f"a {n_sub}-keyframe / {E}-edge 48x64 sub-graph of the bench clip (a full E=276 pass is "
                                      f"estimated at {est_full:.0f} s here), scaled per edge to E=276")
                  + f"; {cores} threads: 1 warm-up + median of 5; 1 thread: one pass over the {n_sub}-keyframe / {E_s}-edge "
                    f"sub-graph, scaled per edge",
    }


# ---------------------------------------------------------------------------------------------- distributed plumbing
class Dist:
    """One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the launcher).  RCCL carries two barriers, the MAX of
    the elapsed time and the one all_gather of the per-clip results; nothing during compute."""

    def __init__(self, use_gpu=True):
        import torch.distributed as dist
        self.dist = dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not use_gpu:  # --mode plumbing
            self.device = torch.device("cpu")
            return
        # the modulo only matters for rehearsing the multi-rank path on a box with fewer GPUs than ranks
        # (VIPE_BENCH_DIST_BACKEND=gloo, ranks share the card)
        self.device = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(self.device)

    def init(self):
        if self.world > 1 and not self.dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = os.environ.get("VIPE_BENCH_DIST_BACKEND", "nccl")  # nccl = RCCL on ROCm
            if backend == "nccl":
                self.dist.init_process_group("nccl", device_id=self.device)
            else:
                self.dist.init_process_group(backend)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        from vipe_amd.driver.clip_shard import exchange_device
        tt = torch.tensor([seconds], device=exchange_device(), dtype=torch.float64)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def n_ranks_seen(self):
        return self.dist.get_world_size() if self.dist.is_initialized() else 1

    def close(self):
        if self.world > 1 and self.dist.is_initialized():
            self.dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------- video clips
def synthetic_frames(device, seed, n_frames, height, width, depth=True):
    """`n_frames` synthetic `Frame`s resident in HBM (decode / resize are host work outside the path): RGB from a pool of
    32 seeded images, the metric depth of a sensor (BASELINE configs[2] "depth_align on": the prior of the dense BA) from a
    pool of 32 seeded 1/U(1,5) depth maps, pinhole intrinsics fx = fy = 0.9 W on the first frame.  No poses: the
    frontend owns them (a clip that carries poses runs the frontend with the motion fixed)."""
    from vipe_amd.slam.system import Frame
    gen = torch.Generator(device="cpu").manual_seed(99 + seed)
    pool_rgb = torch.rand(32, height, width, 3, generator=gen).to(device)
    pool_depth = (1.0 + 4.0 * torch.rand(32, height, width, generator=gen)).to(device) if depth else None
    intr = torch.tensor([0.9 * width, 0.9 * width, width / 2.0, height / 2.0])
    return [Frame(rgb=pool_rgb[f % 32], metric_depth=pool_depth[f % 32] if depth else None,
                  intrinsics=intr if f == 0 else None) for f in range(n_frames)]


def make_clip_runner(device, pipelined=True, height=384, width=512):
    """-> run_clip(seed, n_frames, filter_thresh) -> dict.  Every clip is ONE call of the product's entry point,
    `SLAMSystem.run(frames)` (vipe_amd/slam/system.py, mirror of vipe/slam/system.py:208-316): pass 1 (motion filter,
    keyframes, frontend), the two global-BA passes, pass 2 (every frame through the InnerFiller), the map.  One DroidNet
    (random-init weights, no checkpoint offline) is shared by all clips of this process; every clip gets a fresh system
    (per-clip isolation, run.py:17-26).  Phase times come from `SLAMSystem.timings` (stream drains at the three phase
    boundaries inside run())."""
    from vipe_amd.slam.frontend import FrontendArgs
    from vipe_amd.slam.motion_filter import DroidNet
    from vipe_amd.slam.system import SLAMConfig, SLAMSystem

    torch.manual_seed(0)
    dn = DroidNet()

    def run_clip(seed, n_frames, keep_every=1, frames=None, release_cached_memory=False, backend_lock=None, depth=True):
        from vipe_amd.slam.motion_filter import MotionFilter
        if frames is None:
            frames = synthetic_frames(device, seed, n_frames, height, width, depth=depth)
        # keyframe_thresh = 0: the frontend never drops the second newest keyframe (random-weight flow would otherwise
        # make its distance test drop about half of them and the window would hold ~16 instead of <= 48 edges);
        # filter_thresh = 0: every frame becomes a keyframe (the stress case); keep_every = k: every k-th (scripted_filter)
        cfg = SLAMConfig(buffer=n_frames + 48, filter_thresh=0.0, frontend=FrontendArgs(keyframe_thresh=0.0),
                         pipeline_filter=pipelined, release_cached_memory=release_cached_memory,
                         backend_lock_path=backend_lock)
        sysm = SLAMSystem(device, cfg, droid_net=dn,
                          motion_filter_cls=MotionFilter if keep_every <= 1 else scripted_filter(keep_every))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sysm.run(frames)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = sysm.timings
        traj = out.trajectory.data
        return {"poses": traj.clone(), "intrinsics": out.intrinsics[0, :4].clone(), "frames": n_frames,
                "keyframes": sysm.n_keyframes, "pass1_seconds": tm["pass1_seconds"],
                "seconds_to_global_ba_done": tm["global_ba_done_seconds"], "seconds_to_pass2_done": tm["pass2_done_seconds"],
                "seconds": dt, "finite": bool(torch.isfinite(traj).all()), "update_iterations": sysm.frontend.n_updates,
                "edges_final": int(sysm.frontend.graph.ii.numel()), "backend_edges": sysm.backend_edges,
                "filter_scores": list(sysm.motion_filter.scores), "keep_every": keep_every,
                "backend_lock_wait_seconds": tm.get("backend_lock_wait_seconds"),
                "work_pass1": sysm.work["pass1"], "work": sysm.work["total"]}

    return run_clip


def scripted_filter(keep_every):
    """A MotionFilter whose threshold is scripted per frame so that every `keep_every`-th frame becomes a keyframe: the
    score of a random-weight flow network has no meaningful scale (it barely depends on the frame at all), so a fixed
    threshold keeps either everything or nothing.  ALL of the filter's work still runs for every frame (feature encoder,
    pyramid against the last keyframe, one application of the update operator, the score's read-back); only the
    comparison is replaced.  With a checkpoint the reference's fixed 2.4 px threshold applies."""
    from vipe_amd.slam.motion_filter import MotionFilter

    class ScriptedMotionFilter(MotionFilter):
        def finish(self, h):
            if self.initialized:
                self.thresh = -1.0 if (self.current_frame_idx + 1) % keep_every == 0 else float("inf")
            return super().finish(h)

    return ScriptedMotionFilter


def clip_figures(r):
    return {"slam_system_run": r["frames"] / r["seconds"], "pass1": r["frames"] / r["pass1_seconds"],
            "through_global_ba": r["frames"] / r["seconds_to_global_ba_done"],
            "through_pass2": r["frames"] / r["seconds_to_pass2_done"],
            "frames": r["frames"], "keyframes": r["keyframes"], "update_iterations": r["update_iterations"],
            "backend_edges": r["backend_edges"], "state_finite": r["finite"], "keep_every": r["keep_every"]}


# SURVEY 8(d), per edge at 512x384 (P = 3072): one application of the update iteration moves 4.27 MB (lookup fused into
# the correlation encoder; 6.68 MB unfused) and costs 14.03 GFLOP; one correlation pyramid is 26.7 MB written + 1.57 MB
# read and 2.42 GFLOP; the two encoders are 10.6 GFLOP per frame
MB_PER_EDGE_UPDATE, GF_PER_EDGE_UPDATE = 4.27, 14.03
MB_PER_PYRAMID, GF_PER_PYRAMID, GF_PER_FRAME_ENC = 26.7 + 1.57, 2.42, 10.6


def whole_run_roofline(work, frames, seconds):
    """Algorithmic bytes / FLOPs of a whole clip (host-side edge counts x the per-edge figures above) over its wall time,
    as fractions of the nominal peaks (8 TB/s HBM, 2.5 PFLOP/s dense fp16)."""
    gb = (work["edge_updates"] * MB_PER_EDGE_UPDATE + work["pyramids_built"] * MB_PER_PYRAMID) / 1e3
    tf = (work["edge_updates"] * GF_PER_EDGE_UPDATE + work["pyramids_built"] * GF_PER_PYRAMID + frames * GF_PER_FRAME_ENC) / 1e3
    return {"edge_updates": work["edge_updates"], "pyramids_built": work["pyramids_built"],
            "algorithmic_GB": gb, "algorithmic_TFLOP": tf, "hbm_frac": gb / seconds / 8000.0,
            "mfma_frac": tf / seconds / 2500.0, "peaks": "8 TB/s, 2.5 PFLOP/s"}


def video_mode(args, D):
    """BASELINE config 4 / the frames/s figure of SURVEY 8(d): `--clips` (default: one per rank) independent synthetic
    clips of `--frames` frames, clip i on rank i mod world (clip_shard.run_sharded), each ONE `SLAMSystem.run` (pass 1,
    global BA, pass 2); ONE all_gather of the padded trajectories at the end (RCCL), rank 0 writes the pose / intrinsics
    artifacts.  `--keep-every K`: the motion filter's decision is scripted so that every K-th frame becomes a keyframe
    (default 1: every frame)."""
    from vipe_amd.driver import artifacts
    from vipe_amd.driver.clip_shard import ClipResult, run_sharded

    dev, world, rank = D.device, D.world, D.rank
    D.init()
    run_clip = make_clip_runner(dev, pipelined=not args.no_pipeline, height=args.height, width=args.width)
    run_clip(seed=10_000 + rank, n_frames=24)  # untimed warm-up clip: code objects, workspaces, allocator pools
    from vipe_amd.slam.factor_graph import warm_volume_pool
    warm_volume_pool(dev, 120)  # ... and the memory the 200-keyframe global BA will take (a worker that has run a clip before)
    n_clips = args.clips or world
    stats = []

    def process(cid):
        r = run_clip(seed=cid, n_frames=args.frames, keep_every=max(1, args.keep_every))
        stats.append(r)
        return ClipResult(cid, r["poses"], r["intrinsics"], ok=r["finite"])

    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    results = run_sharded(n_clips, process, f_max=args.frames + 16)
    t_gather_end = time.perf_counter()
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        out_dir = args.out_dir or tempfile.mkdtemp(prefix="vipe_amd_artifacts_")
        written = artifacts.save_clip_results(out_dir, results)
        if not args.out_dir:
            shutil.rmtree(out_dir, ignore_errors=True)
        mine = stats[0] if stats else {}
        frames = n_clips * args.frames
        print(json.dumps({
            "metric": f"frames/s, synthetic {args.width}x{args.height}xN video clips through SLAMSystem.run "
                      f"(pass 1 + global BA + pass 2)",
            "value": frames / dt, "unit": "frames/s", "n_gpus": D.n_ranks_seen(), "steps": frames, "warmup": 24,
            "ms_per_step": 1e3 * dt / max(1, args.frames * ((n_clips + world - 1) // world)),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]-shaped: {n_clips} independent {args.frames}-frame {args.width}x{args.height} clips, "
                                   f"clip-sharded over {world} rank(s), each one SLAMSystem.run (frontend window <= 48 edges, "
                                   f"4+2 update iterations per keyframe, backend.run(7) + backend.run(24), InnerFiller), one "
                                   f"all_gather of the results, artifacts by rank 0",
                       "clips_ok": sum(r.ok for r in results), "clips": len(results), "artifacts_written": len(written),
                       "gather_backend": (D.dist.get_backend() if D.dist.is_initialized() else "none (1 rank)"),
                       "rank0_clip": clip_figures(mine) if mine else None,
                       "rank0_clip_roofline": (whole_run_roofline(mine["work"], 2 * args.frames, mine["seconds"])
                                               if "work" in mine else None),
                       "rank0_seconds_to_gather_end": t_gather_end - t0,
                       "input": "RGB frames + sensor depth resident in HBM: motion filter + feature / context encoders in the "
                                "timed region" + ("" if args.no_pipeline else "; SLAMConfig.pipeline_filter: the filter of "
                                                  "frame f+1 runs on a side stream while the frontend optimises keyframe f")}}))
    D.close()


def clip_worker_mode(args, D):
    """One process = one clip, started by `clips_per_gpu_figure` (K of these share one GPU).  Protocol on stdin / stdout:
    warm-up clip -> "READY" -> wait for a line -> the clip (`SLAMSystem.run`) -> one JSON line (the trajectory goes to
    `--out-dir`/traj_<seed>.npy for the bit-comparison against the K = 1 run)."""
    run_clip = make_clip_runner(D.device, pipelined=not args.no_pipeline, height=args.height, width=args.width)
    run_clip(seed=10_000, n_frames=24)
    if not args.release_cache:
        # a long-lived worker that keeps its pool from clip to clip (one or two per card: their ~134 GB each fit side by
        # side): the global BA's memory is mapped before the start signal, as it is from a worker's second clip on.  Workers
        # that must hand their blocks back (three or more per card) start cold every time - that is what they would see
        from vipe_amd.slam.factor_graph import warm_volume_pool
        warm_volume_pool(D.device, 120)
    frames = synthetic_frames(D.device, args.seed, args.frames, args.height, args.width)  # resident before the start signal
    torch.cuda.synchronize()
    sys.stdout.write("READY\n")
    sys.stdout.flush()
    sys.stdin.readline()
    t0 = time.perf_counter()
    r = run_clip(seed=args.seed, n_frames=args.frames, frames=frames, release_cached_memory=args.release_cache,
                 backend_lock=args.share_card or None, keep_every=max(1, args.keep_every))
    dt = time.perf_counter() - t0
    if args.out_dir:
        np.save(os.path.join(args.out_dir, f"traj_{args.seed}.npy"), r["poses"].cpu().numpy())
    print(json.dumps({"seed": args.seed, "seconds": dt, "frames": r["frames"], "keyframes": r["keyframes"],
                      "pass1_seconds": r["pass1_seconds"], "seconds_to_global_ba_done": r["seconds_to_global_ba_done"],
                      "seconds_to_pass2_done": r["seconds_to_pass2_done"],
                      "backend_lock_wait_seconds": r.get("backend_lock_wait_seconds"),
                      "finite": r["finite"]}))
    sys.stdout.flush()


def clips_per_gpu_figure(args, ks=(1, 2, 4)):
    """K independent clips at once on ONE GPU, one fresh process per clip (clips are independent: the partitioning of
    SURVEY 8e applied below the GPU boundary).  All K workers warm up, then start together; aggregate frames/s = K x
    frames / (release -> last worker's result).  Clip `seed 0` runs at every K: its trajectory is compared with the
    K = 1 run's (the sums the BA accumulates with atomics are order dependent, so equality is reported, not assumed)."""
    out = {}
    tmp = tempfile.mkdtemp(prefix="vipe_amd_kclips_")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # The global BA keeps the correlation pyramids of all its edges (33 MB per edge at 48 x 64: ~3000 edges of a
    # 200-keyframe clip are 99 GB, VIPE_AMD_BACKEND_VOLUME_GB = 160 by default) and fills the chip by itself: the K
    # clips' global-BA phases take turns on a file lock (SLAMConfig.backend_lock_path); what runs concurrently is pass 1 /
    # pass 2 of the other clips.  This function must run in a process that holds no GPU memory itself
    ref = None
    try:
        for K in ks:
            d = os.path.join(tmp, f"k{K}")
            os.makedirs(d)
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--mode", "clip-worker", "--seed", str(k),
                                       "--frames", str(args.frames), "--height", str(args.height), "--width", str(args.width),
                                       "--out-dir", d, "--keep-every", str(max(1, args.keep_every))]
                                      + (["--share-card", os.path.join(d, "backend.lock")] if K > 1 else [])
                                      # two clips' cached backend blocks (2 x ~134 GB) still fit side by side; from three on each
                                      # hands its blocks back after its backend - and the next one pays the driver for scrubbing
                                      # and re-mapping them (~2 s per 100 GB that change owner)
                                      + (["--release-cache"] if K > 2 else []),
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                                      env=env)
                     for k in range(K)]
            try:
                for pr in procs:
                    line = pr.stdout.readline()
                    if line.strip() != "READY":
                        raise RuntimeError(f"worker did not come up: {line!r}")
                t0 = time.perf_counter()
                for pr in procs:
                    pr.stdin.write("go\n")
                    pr.stdin.flush()
                res = [json.loads(pr.stdout.readline()) for pr in procs]
                dt = time.perf_counter() - t0
                for pr in procs:
                    pr.wait(timeout=60)
            finally:
                for pr in procs:
                    if pr.poll() is None:
                        pr.kill()
            traj0 = np.load(os.path.join(d, "traj_0.npy"))
            if ref is None:
                ref = traj0
            out[str(K)] = {"frames_per_s": K * args.frames / dt, "seconds": dt,
                           "pass1_frames_per_s": K * args.frames / max(r["pass1_seconds"] for r in res),
                           "per_clip_seconds": [r["seconds"] for r in res],
                           "backend_lock_wait_seconds": [r.get("backend_lock_wait_seconds") for r in res],
                           "phase_ends_seconds": [[r["pass1_seconds"], r["seconds_to_global_ba_done"], r["seconds_to_pass2_done"]]
                                                  for r in res], "all_finite": all(r["finite"] for r in res),
                           "clip0_trajectory_equal_to_K1": bool(np.array_equal(traj0, ref)),
                           "clip0_max_abs_diff_to_K1": float(np.abs(traj0 - ref).max())}
            _log(f"clips per GPU: K = {K}: {out[str(K)]['frames_per_s']:.1f} frames/s")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    best = max(out, key=lambda k: out[k]["frames_per_s"])
    return {"by_K": out, "best_K": int(best), "frames": args.frames, "keep_every": max(1, args.keep_every),
            "what": "aggregate frames/s of K concurrent SLAMSystem.run clips on one GPU, one process per clip (each with "
                    "its own HIP context and queues), released together after their warm-up; whole run (pass 1 + global BA "
                    "+ pass 2)"}


def backend_mode(args, D):
    """Secondary figure: hot loop B of SURVEY 3.4 - `FactorGraph.update_batch(itrs=2, steps=1)` (reproject, correlation
    volume build + lookup, flow-update operator over all E = 276 edges, one global BA with 2 GN iterations) on the
    headline graph, one clip per GPU."""
    D.init()
    g, buf, graph = build_problem(D.device, args.keyframes, 384, 512, 3, args.extra_edges, seed=1234 + D.rank)

    def step():
        graph.update_batch(itrs=2, steps=1, optimize_intrinsics=False, optimize_rig_rotation=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    if D.rank == 0:
        print(json.dumps({
            "metric": "backend update_batch calls/s, 512x384 48-KF graph", "value": D.world * args.steps / dt,
            "unit": "calls/s", "n_gpus": D.n_ranks_seen(), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": f"update_batch(itrs=2, steps=1), E={int(graph.ii.numel())} edges, "
                                   f"{args.keyframes} keyframes, one clip per GPU"}}))
    D.close()


def plumbing_mode(args, D):
    """The N > 1 plumbing without a GPU (CPU rehearsal of BASELINE config 4's control flow, used by tests/): rank
    processes started by `spawn_ranks`, a gloo process group, `run_sharded` over `--clips` fake clips (seeded
    trajectories, no SLAM), the one all_gather, artifacts by rank 0, one JSON line.  Nothing of the hot path runs here."""
    from vipe_amd.driver import artifacts
    from vipe_amd.driver.clip_shard import ClipResult, run_sharded

    os.environ.setdefault("VIPE_BENCH_DIST_BACKEND", "gloo")
    D.init()
    n_clips = args.clips or D.world

    def process(cid):
        gen = torch.Generator().manual_seed(cid)
        poses = torch.zeros(args.frames, 7)
        poses[:, :3] = torch.randn(args.frames, 3, generator=gen).cumsum(0) * 0.01
        poses[:, 6] = 1.0
        return ClipResult(cid, poses, torch.tensor([460.8, 460.8, 256.0, 192.0]))

    D.barrier()
    t0 = time.perf_counter()
    results = run_sharded(n_clips, process, f_max=args.frames)
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    if D.rank == 0:
        out_dir = args.out_dir or tempfile.mkdtemp(prefix="vipe_amd_artifacts_")
        written = artifacts.save_clip_results(out_dir, results)
        if not args.out_dir:
            shutil.rmtree(out_dir, ignore_errors=True)
        print(json.dumps({"metric": "plumbing only (no GPU work)", "value": n_clips * args.frames / dt, "unit": "frames/s",
                          "n_gpus": D.n_ranks_seen(), "steps": n_clips * args.frames, "warmup": 0, "data": "synthetic",
                          "config": {"workload": "launch + gloo process group + result gather + artifacts",
                                     "clips": len(results), "artifacts_written": len(written),
                                     "clip_ids": [r.clip_id for r in results]}}))
    D.close()


# ---------------------------------------------------------------------------------------------- roofline helper
def profile_traffic(names, flat):
    """HBM traffic of the dominant kernel per launch: FETCH_SIZE x2 + WRITE_SIZE from the builder's own rocprofv3 --pmc
    passes of the bench command (profiles/rNN_summary*.json, committed) - counters cannot be collected from inside the
    timed process, so this is NOT a measurement of the run that prints it.  -> (bytes or None, source or None)"""
    keys = (f"void conv_halo32_kernel<128, 3, true, 0, {'true' if flat else 'false'}>", "void conv_halo32_kernel<128, 3, true, 0>",
            "void conv_halo32_kernel<128, 3, true>")
    for name in names:
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            row = next(prof[k] for k in keys if k in prof)
            return ((row["hbm_read_MB_per_launch"] + row["hbm_write_MB_per_launch"]) * 1e6,
                    f"profiles/{name}: builder-side rocprofv3 --pmc passes of this command, not this run")
        except Exception:  # noqa: BLE001
            pass
    return None, None


def conv_roofline(graph, step, device, prof_steps):
    """Roofline of the dominant kernel: conv_halo32_kernel<128, 3, true, 0, FLAT> (every 3x3 convolution of the flow-update
    operator with >= 128 output channels: corr2, z|r, q, delta0|weight0|agg1, agg2 - 5 launches per step, ~50 % of the
    step).  Every launch of that instantiation in `prof_steps` extra steps is bracketed by events on the launch stream;
    achieved = (algorithmic flops of those launches) / (their summed duration).
    -> (avg launch ms, flops per launch, achieved TFLOP/s, launches per step)"""
    eng = graph.update_op.engine(device)
    rec = []
    recording = False
    orig = eng._conv

    def timed_conv(pk, x0, x0_coff, B, H, W, *a, **k):
        cin = k.get("cin") or pk.cin
        # the dominant instantiation only (<128, 3, true, 0>): the z|r convolution that starts from the staged fp32 partial
        # sums is its own instantiation (<..., 1>) and rocprof row
        staged = k.get("accinit") is not None and k["accinit"].dtype == torch.float32
        if recording and pk.cout > 64 and pk.kh == 3 and not staged:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(pk, x0, x0_coff, B, H, W, *a, **k)
            e1.record()
            rec.append((e0, e1, 2.0 * B * H * W * cin * pk.cout * pk.kh * pk.kw))
        else:
            orig(pk, x0, x0_coff, B, H, W, *a, **k)

    # the operator is normally ONE natively sequenced library call; for these few steps the same kernels are issued one
    # by one from Python (forward_nhwc(native=False)) so that single launches can be bracketed by events
    fwd = eng.forward_nhwc
    eng._conv, eng.forward_nhwc = timed_conv, (lambda *a, **k: fwd(*a, **dict(k, native=False)))
    try:
        # three unrecorded steps first: the gather / host work after the timed region lets the clocks fall, and the first
        # kernels after an idle gap run 10-50 % slower than in the timed region (kernel trace of this very command)
        for _ in range(3 if prof_steps else 0):
            step()
        recording = True
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize()
    finally:
        eng._conv = orig
        del eng.forward_nhwc
    tot_ms = sum(a.elapsed_time(b) for a, b, _ in rec) or float("nan")
    tot_fl = sum(f for _, _, f in rec)
    return (tot_ms / max(1, len(rec)), tot_fl / max(1, len(rec)), tot_fl / (tot_ms * 1e-3) / 1e12,
            len(rec) // max(1, prof_steps))


def profile_counter_bytes(substr, names=("r04_summary.json", "r03_summary.json")):
    """HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE) of the kernel whose summary key contains `substr`, from the
    builder's rocprofv3 --pmc passes of the bench command (profiles/rNN_summary.json) -> (bytes or None, source or None)"""
    for name in names:
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", name)))
            row = next(v for k, v in prof.items() if substr in k and "hbm_read_MB_per_launch" in v)
            return ((row["hbm_read_MB_per_launch"] + row.get("hbm_write_MB_per_launch", 0.0)) * 1e6,
                    f"profiles/{name} (builder-side --pmc passes of this command, not this run)")
        except Exception:  # noqa: BLE001
            pass
    return None, None


def hbm_kernel_rooflines(graph, step, device, n=8):
    """The three HBM-bound kernels north_star names, each as event-timed launches on this run's own data, against the
    algorithmic bytes of SURVEY 8(d) and the 8 TB/s nominal peak; `traffic` = the counter bytes of the builder's --pmc
    passes.  (i) fused lookup (`corr_lookup_conv_kernel`: 4-level 7x7 lookup + the correlation encoder's 1x1, launched
    alone on the step's own pyramid / coordinates); (ii) BA accumulate (`ba_accum_mfma_kernel`, the last Gauss-Newton
    iteration's launch inside real eager steps, bracketed by events the library records: vipe_ba_params.profile_ev0/1 -
    it runs beside the staged gate convolution there, as in the timed region); (iii) pyramid build
    (`corr_pyramid_build_kernel`, all E edges rebuilt into their own pool slots)."""
    from vipe_amd.ext import droid_net_ext, slam_ext
    E = int(graph.ii.numel())
    P = graph.ht * graph.wd
    eng = graph.update_op.engine(device)
    out = {}

    def entry(kernel, ms, alg_bytes, key, what):
        traffic, src = profile_counter_bytes(key)
        ach = alg_bytes / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "bound": "hbm", "avg_launch_ms": ms, "algorithmic_bytes": alg_bytes, "achieved": ach,
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": traffic,
                "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None, "traffic_source": src, "what": what}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / n

    # (i) fused lookup
    plan = graph._edge_plan()
    corr = graph.corr.lookup_deferred(plan["io"][0])
    if isinstance(corr, tuple):
        c1 = torch.empty((E, graph.ht, graph.wd, 128), dtype=torch.float16, device=device)
        ms = timed(lambda: droid_net_ext.corr_lookup_conv1x1(corr[1], corr[2], eng.corr0.packed, eng.corr0.bias, c1, act="relu",
                                                             slots=corr[3], grid=corr[4]))
        per_edge = P * (4 * 64 * 2 + 8 + 128 * 2)  # 8 x 8 taps x 4 levels x 2 B read + coords + 128 fp16 channels written
        out["corr_lookup_conv_kernel"] = entry("corr_lookup_conv_kernel<true>", ms, E * per_edge, "corr_lookup_conv_kernel",
                                               "1.573 MB taps + 0.025 MB coords + 0.786 MB written per edge")
    # (ii) BA accumulate: the library records the two events around the last iteration's launch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    torch.cuda.synchronize()
    try:
        tot, per_iter = 0.0, [0.0, 0.0, 0.0]
        step()
        for k in range(3 * n):  # every Gauss-Newton iteration's launch in turn: each shares the chip with another piece of the
            slam_ext.PROFILE_EVENTS = (e0, e1, k % 3)  # staged gate convolution (the first with none of it)
            step()
            torch.cuda.synchronize()
            per_iter[k % 3] += e0.elapsed_time(e1) / n
        tot = sum(per_iter) / 3.0
    finally:
        slam_ext.PROFILE_EVENTS = None
    out["ba_accum_mfma_kernel"] = entry("ba_accum_mfma_kernel<0,0>", tot, E * P * (2 * 8 + 4) + 48 * P * 16,
                                        "ba_accum_mfma_kernel",
                                        "0.061 MB per edge and iteration (target, weight, disparity) + 4 node-side maps per "
                                        "keyframe; latency- / reduction-bound by construction (SURVEY F5); avg_launch_ms = mean over "
                                        "the three Gauss-Newton iterations' launches")
    out["ba_accum_mfma_kernel"]["avg_launch_ms_per_gn_iteration"] = per_iter
    # (iii) pyramid build of all edges, into the slots they already own
    V = graph.buffer.n_views
    pi, qi, pj, qj = (plan[k] for k in ("pi", "qi", "pj", "qj"))
    f1, f2 = (pi, pj) if V == 1 else (pi * V + qi, pj * V + qj)
    fm = graph.buffer.flattened_fmaps
    if getattr(graph.corr, "blocked", False):
        ms = timed(lambda: droid_net_ext.corr_pyramid_build_indexed(fm, f1.contiguous(), f2.contiguous(), levels=graph.corr.pool,
                                                                    slots=graph.corr.slots,
                                                                    frame_range=(0, graph.buffer.n_frames * V)))
        out["corr_pyramid_build_kernel"] = entry("corr_pyramid_build_kernel<true>", ms,
                                                 E * (2 * 128 * P * 2 + int(P * P * 2 * (1 + 1 / 4 + 1 / 16 + 1 / 64))),
                                                 "corr_pyramid_build_kernel", "2 x 0.786 MB maps read + 25.1 MB pyramid written per edge")
    return out


def capture_two_steps(step):
    """two consecutive update iterations as one HIP graph (see update_mode); None when capture is not possible"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # allocate everything the two steps need outside the capture
        step()
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    # thread_local: the RCCL watchdog thread of a multi-GPU run may touch the runtime while this thread captures
    with torch.cuda.graph(cg, capture_error_mode="thread_local"):
        step()
        step()
    torch.cuda.synchronize()
    cg.replay()  # one untimed replay
    return cg


def grid_figure(device, height, width, n_kf, steps, prof_steps, headline_px_rate=None):
    """The headline measurement (graph replay of two captured update iterations, radius-3 graph, depth prior on, 3 GN
    iterations) on another image size, with the event-timed roofline of its dominant convolution."""
    g, buf, graph = build_problem(device, n_kf, height, width, 3, 0, seed=1234)
    E = int(graph.ii.numel())

    def step():
        graph.update(t0=1, t1=n_kf, itrs=3)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    try:
        cg, launch = capture_two_steps(step), "hipgraph (2 steps per replay)"
    except Exception as e:  # noqa: BLE001
        cg, launch = None, f"eager (graph capture failed: {type(e).__name__})"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // 2):
        cg.replay() if cg is not None else (step(), step())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = 2 * (steps // 2)
    finite = bool(torch.isfinite(buf.poses[:n_kf]).all() and torch.isfinite(buf.disps[:n_kf]).all()
                  and torch.isfinite(graph.target).all())
    ms, fpl, ach, lps = conv_roofline(graph, step, device, prof_steps)
    P = g.ht * g.wd
    flat = not (g.wd % 64 == 0 and g.ht % 4 == 0)
    traffic, traffic_src = profile_traffic(("r03_summary_41x73.json",), True) if (g.ht, g.wd) == (41, 73) else (None, None)
    out = {"value": n / dt, "unit": "iters/s", "ms_per_step": 1e3 * dt / n, "steps": n, "grid": [g.ht, g.wd], "pixels": P,
           "keyframes": n_kf, "edges": E, "launch": launch, "state_finite": finite,
           "edge_pixels_per_s": n / dt * E * P,
           "tile_kernels": {"gate_context_hoisted": graph.pgate is not None,
                            "pyramid_store": "blocked" if getattr(graph.corr, "blocked", False) else "reference",
                            "staged_gates": getattr(graph, "_gate_state", None) is not None},
           "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_FP16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": ms,
                        "flops_per_launch": fpl, "launches_per_step": lps,
                        "kernel": f"conv_halo32_kernel<128, 3, true, 0, {'true' if flat else 'false'}> on this grid "
                                  f"({'flat' if flat else '4 x 64'} tiling; event-timed launches)"}}
    if headline_px_rate:
        out["per_pixel_throughput_vs_48x64"] = out["edge_pixels_per_s"] / headline_px_rate
    del graph, buf
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------- secondary figures
def secondary_figures(args, device, graph, step, headline_px_rate=None):
    """Figures north_star asks to be reported next to the headline, measured in the same driver-run process AFTER (and
    outside) the timed headline region, N = 1 only.  Each is a plain eager-launch timing."""
    out = {}

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)

    # (1) the same step with ALL of the operator's work inside the iteration: the context-feature part of the GRU gates
    # recomputed every iteration (as the reference does) instead of once per edge at add_factors
    if getattr(graph, "pgate", None) is not None:
        keep, graph.pgate = graph.pgate, None
        out["value_all_gate_work_per_iteration"] = timed(step, 6)
        graph.pgate = keep
    _log("secondary: all-gate-work done")
    # (1b) the same measurement on the grid the reference's resize produces for 16:9 video (1280 x 720 -> 328 x 584 ->
    # 41 x 73, vipe/slam/system.py:46-59; both clips of assets/examples), and on BASELINE configs[4]'s 1024 x 512 (64 x 128)
    for key, (hh, ww, nk, st) in {"value_grid_41x73": (328, 584, args.keyframes, 20),
                                  "value_config5_grid_64x128": (512, 1024, args.keyframes, 10)}.items():
        try:
            out[key] = grid_figure(device, hh, ww, nk, st, args.prof_steps, headline_px_rate)
        except Exception as e:  # noqa: BLE001
            out[key] = f"failed: {type(e).__name__}: {e}"
        _log(f"secondary: {key} done")
    # (2) E = 768 stress (backend cap 16 t): radius-3 graph + 492 seeded long-range edges
    try:
        _, _, g768 = build_problem(device, args.keyframes, 384, 512, 3, 492, seed=4321)
        out["value_E768"] = timed(lambda: g768.update(t0=1, t1=args.keyframes, itrs=3), 5)
        out["E768_edges"] = int(g768.ii.numel())
        del g768
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001 - a secondary figure must not take the headline down with it
        out["value_E768"] = f"failed: {type(e).__name__}: {e}"
    _log("secondary: E768 done")
    # (2b) hot loop B (factor_graph.py:317-394): `update_batch(itrs=2, steps=1)` on the headline graph - reproject, correlation
    # of all E edges, the operator, one global BA with 2 GN iterations - on the volume path (pyramids rebuilt per call: steps = 1
    # keeps nothing) and on the reference's own volume-free AltCorr path
    try:
        _, _, gb = build_problem(device, args.keyframes, 384, 512, 3, 0, seed=1234)
        call = lambda: gb.update_batch(itrs=2, steps=1, optimize_intrinsics=False, optimize_rig_rotation=False)  # noqa: E731
        ub = {"volume_path": timed(call, 5)}
        os.environ["VIPE_AMD_BACKEND_ALTCORR"] = "1"
        try:
            ub["altcorr_path"] = timed(call, 5)
        finally:
            del os.environ["VIPE_AMD_BACKEND_ALTCORR"]
        ub["edges"] = int(gb.ii.numel())
        out["update_batch_calls_per_s"] = ub
        del gb
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        out["update_batch_calls_per_s"] = f"failed: {type(e).__name__}: {e}"
    _log("secondary: update_batch done")
    # (3) synthetic video: the product's entry point, SLAMSystem.run (pass 1, the two global-BA passes, pass 2, the map),
    # on 200 RGB + depth frames resident in HBM - every frame a keyframe (stress case), then the same with the motion
    # filter's threshold scripted to keep about one frame in four (400 frames)
    try:
        from vipe_amd.slam.factor_graph import warm_volume_pool
        run_clip = make_clip_runner(device)
        run_clip(seed=10_000, n_frames=24)
        # the state of a worker that has run a clip of this size before: the global BA's ~100 GB already mapped, in one block
        # (whatever this process cached for other shapes is handed back first)
        torch.cuda.empty_cache()
        warm_volume_pool(device, 120)
        r = run_clip(seed=0, n_frames=args.frames)
        fps = clip_figures(r)
        fps.update({
            "roofline_pass1": whole_run_roofline(r["work_pass1"], r["frames"], r["pass1_seconds"]),
            "roofline_whole_clip": whole_run_roofline(r["work"], 2 * r["frames"], r["seconds"]),
            "what": "one synthetic 512x384 clip = ONE SLAMSystem.run(frames) call, RGB + sensor-depth frames resident in HBM, "
                    "every frame a keyframe: slam_system_run = frames / wall time of the call (pass 1: motion filter + encoders "
                    "+ proximity edges + 4+2 update iterations per keyframe incl. the 8-keyframe initialisation, the filter of "
                    "frame f+1 on a side stream under keyframe f's frontend step; backend.run(7) + backend.run(24); pass 2: "
                    "every frame encoded again, chunks of 16 through the InnerFiller; extract_slam_map); pass1 / "
                    "through_global_ba / through_pass2 = the same clip up to that phase boundary (SLAMSystem.timings)"})
        r0 = run_clip(seed=0, n_frames=args.frames, depth=False)
        fps["without_sensor_depth"] = dict(clip_figures(r0),
                                           what="the same clip without sensor depth on the frames - the conditions of round 3's "
                                                "hand-driven clip (117.7 frontend only / 73.3 with global BA / 65.5 with infill in "
                                                "BENCH_r03): without the prior the random-weight system drifts further, the global "
                                                "BA's proximity graph holds ~1700 instead of ~3000 edges")
        r4 = run_clip(seed=1, n_frames=2 * args.frames, keep_every=4)
        fps["keep_one_in_four"] = dict(clip_figures(r4), keep_rate=r4["keyframes"] / r4["frames"],
                                       what="the same with the filter's decision scripted to keep every 4th frame "
                                            "(bench.scripted_filter): non-keyframes cost one filter check in pass 1 and one "
                                            "InnerFiller slot in pass 2")
        out["frames_per_s"] = fps
    except Exception as e:  # noqa: BLE001
        out["frames_per_s"] = f"failed: {type(e).__name__}: {e}"
    _log("secondary: video done")
    # (4) the same clip at the size the reference's resize gives 16:9 video (1280 x 720 -> 584 x 328, 41 x 73 grid): what
    # BASELINE configs[1] (assets/examples) actually runs at
    try:
        run169 = make_clip_runner(device, height=328, width=584)
        run169(seed=10_000, n_frames=24)
        torch.cuda.empty_cache()
        warm_volume_pool(device, 120)
        r = run169(seed=0, n_frames=args.frames)
        out["frames_per_s_584x328"] = dict(clip_figures(r), grid=[41, 73])
    except Exception as e:  # noqa: BLE001
        out["frames_per_s_584x328"] = f"failed: {type(e).__name__}: {e}"
    _log("secondary: 16:9 video done")
    return out


def iteration_roofline(graph, E, n_kf, P, steps_per_s, device):
    """The whole update iteration against both nominal peaks, per GPU.  `mfma_frac_executed` counts the FLOPs this build
    EXECUTES per iteration; `mfma_frac_reference_equivalent` counts the reference's (14.03 GFLOP per edge +
    1.37 per source node), i.e. it also credits the context-feature part of the GRU gates - conv3x3(inp; W_{z|r|q}[:,
    128:256]), constant per edge like the correlation volume - which is computed once at add_factors and NOT per iteration."""
    eng = graph.update_op.engine(device)
    ref_tf = (E * GF_PER_EDGE_UPDATE + n_kf * 1.37) / 1e3
    hoisted_tf = (E * P * eng.gates_inp.flops_per_pixel / 1e12) if getattr(graph, "pgate", None) is not None else 0.0
    gb = E * MB_PER_EDGE_UPDATE / 1e3
    return {"algorithmic_GB": gb, "hbm_frac": gb * steps_per_s / PEAK_HBM_GBS,
            "executed_TFLOP": ref_tf - hoisted_tf, "mfma_frac_executed": (ref_tf - hoisted_tf) * steps_per_s / PEAK_FP16_TFLOPS,
            "reference_equivalent_TFLOP": ref_tf, "mfma_frac_reference_equivalent": ref_tf * steps_per_s / PEAK_FP16_TFLOPS,
            "hoisted_gate_context_TFLOP": hoisted_tf,
            "what": "one update iteration over the measured step time, per GPU; executed = reference-equivalent minus the "
                    "gate-context convolution hoisted out of the iteration (computed once per edge)"}


# ---------------------------------------------------------------------------------------------- headline
def update_mode(args, D):
    from vipe_amd.driver.clip_shard import ClipResult, gather_results

    device, world, rank = D.device, D.world, D.rank
    # clip sharding: rank r owns clip r (seed differs per rank), no exchange during compute
    g, buf, graph = build_problem(device, args.keyframes, args.height, args.width, 3, args.extra_edges, seed=1234 + rank)
    E = int(graph.ii.numel())
    if args.serial_operator:  # profiling aid: every kernel of the operator on ONE stream (per-kernel durations then are
        graph.update_op.engine(device).op_side_min_edges = 10 ** 9  # those of the kernel running alone)

    def step():
        graph.update(t0=1, t1=args.keyframes, itrs=3)

    _log(f"problem built: E = {E}")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    _log("warm-up done")

    # The update iteration has no host read-back and no shape that changes from step to step, so TWO consecutive steps
    # (the hidden state ping-pongs between two buffers; targets / weights / poses / disparities are updated in place, so
    # every replay continues from the state the previous one left) are captured once into a HIP graph and the timed
    # region replays it: the same kernels on the same stream, minus the ~45 Python / ctypes launches per step - which
    # keeps the figure independent of host jitter (eight ranks share one host's cores in the multi-GPU run).
    # `--no-hipgraph` times eager launches.
    launch, cg = "eager", None
    if not args.no_hipgraph and args.steps >= 2:
        try:
            cg = capture_two_steps(step)
            launch = "hipgraph (2 steps per replay)"
        except Exception as e:  # noqa: BLE001 - fall back to eager launches and say so
            cg = None
            launch = f"eager (graph capture failed: {type(e).__name__})"
    # RCCL comes up only now: problem set-up, warm-up and the graph capture above are rank-local (clip sharding has no
    # data-path collective), so no communicator thread is alive while a stream is being captured
    _log(f"launch mode: {launch}")
    D.init()
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    if cg is not None:
        for _ in range(args.steps // 2):
            cg.replay()
        if args.steps % 2:
            step()
    else:
        for _ in range(args.steps):
            step()
    torch.cuda.synchronize()
    t_steps = time.perf_counter() - t0
    # the job's one exchange: every rank's clip result (trajectory + intrinsics, ~1.4 KB here) in one all_gather
    mine = ClipResult(rank, buf.poses[:args.keyframes], buf.intrinsics[0, :4], ok=True, seconds=t_steps)
    results = gather_results([mine], n_clips=world, f_max=args.keyframes)
    gather_ms = 1e3 * (time.perf_counter() - t0 - t_steps)
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    finite = bool(torch.isfinite(buf.poses[:args.keyframes]).all() and torch.isfinite(buf.disps[:args.keyframes]).all()
                  and torch.isfinite(graph.target).all())
    n_seen = D.n_ranks_seen()
    _log(f"timed region done: {world * args.steps / dt:.1f} it/s")

    gate_ms, flops_per_launch, achieved, launches_per_step = conv_roofline(graph, step, device, args.prof_steps)

    # HBM traffic of the dominant kernel per launch: FETCH_SIZE x2 + WRITE_SIZE from the builder's own rocprofv3 --pmc
    # passes of this command (profiles/rNN_summary.json, committed) - counters cannot be collected from inside the
    # timed process, so this is NOT a measurement of this run
    traffic, traffic_src = profile_traffic(("r03_summary.json", "r02_summary.json", "r01_summary.json"), False)

    # matrix-core utilisation of the same kernel from the builder's SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass
    # (profiles/r03_mfma_util.json) - like `traffic`, evidence collected beside this run, not by it
    mfma_pmc = None
    for name in ("r03_mfma_util.json", "r02_mfma_util.json"):
        try:
            mu = json.load(open(os.path.join(ROOT, "profiles", name)))
            mu = mu.get("void conv_halo32_kernel<128, 3, true, 0, false>") or mu["void conv_halo32_kernel<128, 3, true, 0>"]
            mfma_pmc = {"mfma_busy_frac": mu["mfma_util"], "effective_clock_GHz": mu["effective_clock_GHz"],
                        "avg_launch_ms": mu["avg_us"] / 1e3, "source": f"profiles/{name} (builder-side --pmc pass)"}
            break
        except Exception:  # noqa: BLE001
            pass

    if rank == 0:
        out = {
            "metric": "dense-BA+flow update iters/s, 512x384 48-KF graph",
            "value": world * args.steps / dt,
            "unit": "iters/s",
            "n_gpus": n_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": f"{'configs[2]: ' if (args.height, args.width) == (384, 512) else 'NOT the headline size: '}"
                                   f"{args.width}x{args.height}, {args.keyframes}-keyframe factor graph, E={E} edges "
                                   f"(radius-3 bidirectional), depth_align on (sensor-depth prior on every keyframe), "
                                   f"3 GN iterations per update, one clip per GPU",
                       "parallelism": f"clip-sharded x{world}", "launch": launch,
                       "result_gather": {"clips": len(results), "ms": gather_ms, "inside_timed_region": True,
                                         "backend": D.dist.get_backend() if D.dist.is_initialized() else "none (1 rank)"},
                       "state_finite": finite,
                       "gate_overlap": (f"hidden-state part of the next iteration's z|r gates + global context on a second "
                                        f"stream under the BA ({graph.gate_overlap_mode})"
                                        if getattr(graph, "_gate_state", None) is not None else "off"),
                       "operator_streams": 1 if args.serial_operator or E < graph.update_op.engine(device).op_side_min_edges else 2,
                       "gate_context": "context-feature part of the GRU gates computed once per edge (at add_factors, "
                                       "like the correlation volume), not per iteration"
                                       if getattr(graph, "pgate", None) is not None else "recomputed every iteration"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP16_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "conv_halo32_kernel<128, 3, true, 0> (NHWC fp16 implicit-GEMM 3x3 conv on MFMA 16x16x32: the "
                                   "launches with Cout >= 128 of the flow-update operator on the main stream, except the "
                                   "z|r convolution that starts from staged fp32 partial sums = instantiation <..., 1>)",
                         "avg_launch_ms": gate_ms, "flops_per_launch": flops_per_launch,
                         "launches_per_step": launches_per_step, "pmc": mfma_pmc},
            # the whole update iteration against both nominal peaks (SURVEY 8d totals per edge; N source nodes add
            # 1.37 GFLOP each): MFMA-bound by construction, the HBM figure is what north_star asks to see beside it
            "iteration_roofline": iteration_roofline(graph, E, args.keyframes, g.ht * g.wd, args.steps / dt / world, device),
        }
        if world == 1:
            try:
                out["roofline_hbm_kernels"] = hbm_kernel_rooflines(graph, step, device)
            except Exception as e:  # noqa: BLE001
                out["roofline_hbm_kernels"] = f"failed: {type(e).__name__}: {e}"
            _log("HBM-bound kernels timed")
        if world == 1 and not args.no_secondary:
            out.update(secondary_figures(args, device, graph, step, out["value"] * E * g.ht * g.wd))
            if isinstance(out.get("frames_per_s"), dict):
                out["frames_per_s"]["clips_per_gpu"] = args.kclips
        out["cpu_baseline"] = cpu_baseline() if (world == 1 and not args.no_cpu_baseline) else None
        print(json.dumps(out))
    D.close()
    if not finite:
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--keyframes", type=int, default=48)
    ap.add_argument("--height", type=int, default=384, help="update mode: image height (default: the headline's 384)")
    ap.add_argument("--width", type=int, default=512, help="update mode: image width (328 x 584 = the 41 x 73 grid of 16:9 video)")
    ap.add_argument("--extra-edges", type=int, default=0, help="seeded long-range edges on top of the radius-3 graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="update mode: skip the secondary figures (all-gate-work it/s, E=768 it/s, video frames/s)")
    ap.add_argument("--prof-steps", type=int, default=4)
    ap.add_argument("--serial-operator", action="store_true",
                    help="do not use the operator's second stream (vipe_update_buffers.side_stream): for kernel-stats profiles")
    ap.add_argument("--mode", default="update", choices=["update", "video", "backend", "plumbing", "clip-worker", "clips-per-gpu"],
                    help="update: the headline metric (update iterations/s on the 48-keyframe graph); video: frames/s "
                         "of independent synthetic clips through the keyframe frontend, clip-sharded over the ranks "
                         "(BASELINE config 4); backend: FactorGraph.update_batch calls/s (hot loop B) on the same graph")
    ap.add_argument("--frames", type=int, default=200, help="frames per synthetic clip")
    ap.add_argument("--clips", type=int, default=0, help="video mode: number of clips (default: one per rank)")
    ap.add_argument("--out-dir", default=None, help="video mode: keep rank 0's pose / intrinsics artifacts here")
    ap.add_argument("--no-hipgraph", action="store_true",
                    help="update mode: time eager launches instead of replaying the captured two-step HIP graph")
    ap.add_argument("--keep-every", type=int, default=1,
                    help="video mode: script the motion filter so that every K-th frame becomes a keyframe (1: every frame)")
    ap.add_argument("--seed", type=int, default=0, help="clip-worker mode: the clip's seed")
    ap.add_argument("--release-cache", action="store_true",
                    help="clip-worker mode: SLAMConfig.release_cached_memory (more than two clips on one card: their cached "
                         "pyramid blocks would not fit side by side)")
    ap.add_argument("--share-card", default="",
                    help="clip-worker mode: other clips run on this GPU - path of the lock file their global-BA phases "
                         "take turns on (SLAMConfig.backend_lock_path)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="video mode: SLAMConfig.pipeline_filter = False (the motion filter of every frame on the main stream, "
                         "strictly before the frontend step)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args.gpus))  # nothing above touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        sys.exit(2)

    if args.mode == "clips-per-gpu":  # the parent only starts and times the workers: it never touches the GPU
        print(json.dumps(clips_per_gpu_figure(args)))
        return
    kclips = None
    if args.mode == "update" and world == 1 and not args.no_secondary:
        # K clips per GPU, one process per clip: measured FIRST, while this process has not touched the GPU (its own
        # problems and caches - tens of GB later on - would take the memory two concurrent global BAs need)
        try:
            kclips = clips_per_gpu_figure(args)
        except Exception as e:  # noqa: BLE001 - a secondary figure must not take the headline down with it
            kclips = f"failed: {type(e).__name__}: {e}"
        _log("clips per GPU done")
    args.kclips = kclips
    D = Dist(use_gpu=args.mode != "plumbing")
    {"update": update_mode, "video": video_mode, "backend": backend_mode, "plumbing": plumbing_mode,
     "clip-worker": clip_worker_mode}[args.mode](args, D)


if __name__ == "__main__":
    main()
