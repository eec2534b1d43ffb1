"""bench.py - dense-BA + flow update iterations per second on the BASELINE.json workload.

One "step" = one FactorGraph.update: reproject -> 4-level correlation lookup -> flow-update
operator (ConvGRU) -> dense bundle adjustment (3 Gauss-Newton iterations) on a synthetic 512x384 clip,
48 keyframes, E = 276 edges (radius-3 bidirectional graph), inputs resident in HBM.  With --gpus N every
rank runs its own clip (clip sharding, no data-path collective); the value is the whole-job aggregate.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0  # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0


def build_problem(device, n_kf, height, width, radius, extra_edges, conv_backend, seed=1234):
    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.factor_graph import FactorGraph
    from vipe_amd.slam.networks import UpdateModule
    from vipe_amd.synth import make_graph

    os.environ["VIPE_AMD_CONV"] = conv_backend
    g = make_graph(n=n_kf, height=height, width=width, radius=radius, extra_edges=extra_edges, seed=seed)
    buf = GraphBuffer(height, width, n_views=1, buffer_size=max(64, n_kf), device=device)
    buf.n_frames = n_kf
    buf.poses[:n_kf] = torch.from_numpy(g.poses).to(device)
    buf.disps[:n_kf, 0] = torch.from_numpy(g.disps).to(device)
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


def conv_flops(E, ht, wd, cin, cout, k):
    return 2.0 * E * ht * wd * cin * cout * k * k


def cpu_baseline(seconds_budget=20.0):
    """Oracle (CPU restatement, 'port') of corr lookup + dense BA on a bounded sub-graph, 1 thread."""
    from threadpoolctl import threadpool_limits

    from oracle import ba as oba
    from oracle import corr as ocorr
    from oracle import se3 as ose3
    from vipe_amd.synth import make_graph

    n_sub = 32
    g = make_graph(n=n_sub, height=384, width=512, radius=3, seed=1234)
    E = len(g.ii)
    rng = np.random.default_rng(0)
    # one pyramid for all sample edges (random fp16 volume: the lookup cost does not depend on the values)
    levels = [np.broadcast_to(rng.normal(0, 1, (1, g.ht, g.wd, g.ht >> i, g.wd >> i)).astype(np.float16),
                              (E, g.ht, g.wd, g.ht >> i, g.wd >> i)) for i in range(4)]
    coords = np.stack(np.meshgrid(np.arange(g.wd), np.arange(g.ht)), -1).astype(np.float32)[None, None].repeat(E, 1)
    coords = coords + rng.normal(0, 2, coords.shape).astype(np.float32)
    with threadpool_limits(limits=1):
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        ocorr.corr_lookup(levels, coords, 3)
        t_corr = time.perf_counter() - t0
        t0 = time.perf_counter()
        oba.bundle_adjustment(g.poses, g.disps[:, None], g.disps_sens[:, None], g.intrinsics, ose3.se3_identity(1),
                              g.target.reshape(E, -1, 2), g.weight.reshape(E, -1, 2), g.eta[:, None], g.ii, g.jj, t0=1,
                              t1=n_sub, n_iters=3, pose_damping=1e-3, pose_ep=0.1, dtype=np.float32)
        t_ba = time.perf_counter() - t0
    per_edge = (t_corr + t_ba) / E
    return {
        "value": 1.0 / (per_edge * 276), "unit": "iters/s", "cores": 1, "kind": "port",
        "sample": f"oracle corr lookup + 3-iteration dense BA (GRU excluded) on a {n_sub}-keyframe / {E}-edge "
                  f"48x64 sub-graph ({t_corr:.1f}s + {t_ba:.1f}s), scaled per edge to E=276",
    }


def backend_mode(args, device, world, rank):
    """Secondary figure: hot loop B of SURVEY 3.4 - `FactorGraph.update_batch(itrs=2, steps=1)` (reproject, correlation
    volume build + lookup, flow-update operator over all E = 276 edges, one global BA with 2 GN iterations) on the
    headline graph, one clip per GPU."""
    import torch.distributed as dist

    g, buf, graph = build_problem(device, args.keyframes, 384, 512, 3, args.extra_edges, args.conv, seed=1234 + rank)

    def step():
        graph.update_batch(itrs=2, steps=1, optimize_intrinsics=False, optimize_rig_rotation=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({
            "metric": "backend update_batch calls/s, 512x384 48-KF graph", "value": world * args.steps / dt,
            "unit": "calls/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 (correlation, GRU) + f32 geometry/BA", "data": "synthetic",
            "config": {"workload": f"update_batch(itrs=2, steps=1), E={int(graph.ii.numel())} edges, "
                                   f"{args.keyframes} keyframes, one clip per GPU"}}))


def video_mode(args, device, world, rank):
    """Secondary figure of SURVEY 8(d): frames/s of the keyframe frontend (frontend.py:78-167 mirror) on a synthetic
    512x384xN "video" in which every frame is a keyframe: per frame = motion filter on the RGB frame (feature encoder,
    one flow-update application against the last keyframe, context encoder), proximity-edge proposal (frame_distance kernel),
    correlation volume + pyramid + gate-context build for the new edges, 4 (+2) update iterations over the <= 48-edge
    window incl. the dense BA with inactive edges.  Frames are seeded uniform RGB images resident in HBM
    (`--video-features`: seeded N(0,1) feature maps instead, encoders skipped), poses follow a smooth seeded trajectory (constant-velocity initialisation, then the BA moves
    them), random-init operator weights."""
    import torch.distributed as dist

    from vipe_amd.slam.buffer import GraphBuffer
    from vipe_amd.slam.frontend import FrontendArgs, SLAMFrontend
    from vipe_amd.slam.motion_filter import DroidNet, MotionFilter

    N = args.frames
    torch.manual_seed(1234 + rank)
    buf = GraphBuffer(384, 512, buffer_size=N + 16, device=device)
    buf.intrinsics[:] = torch.tensor([460.8, 460.8, 256.0, 192.0], device=device)
    torch.manual_seed(0)
    dn = DroidNet()  # fnet + cnet + update operator, random-init weights (no checkpoint offline)
    um = dn.update
    # keyframe_thresh = 0: every synthetic frame stays a keyframe (random-weight flow would otherwise make the
    # distance test drop about half of them and the window would hold ~16 instead of <= 48 edges)
    fe = SLAMFrontend(um, buf, FrontendArgs(keyframe_thresh=0.0), device)
    gen = torch.Generator(device="cpu").manual_seed(99 + rank)
    pool_d = (1.0 / (1.0 + 4.0 * torch.rand(32, 48, 64, generator=gen))).to(device)
    if args.video_features:
        # legacy variant: the "decoded + encoded" frames are a pool of seeded feature maps resident in HBM
        pool_f = torch.randn(32, 128, 48, 64, generator=gen).half().to(device)
        pool_n = torch.randn(32, 128, 48, 64, generator=gen).tanh().half().to(device)
        pool_i = torch.randn(32, 128, 48, 64, generator=gen).relu().half().to(device)
    else:
        # decoded RGB frames resident in HBM before the timed region (decode / resize are host work outside the path);
        # every frame goes through the motion filter: feature encoder, one flow-update application against the last
        # keyframe, context encoder (thresh 0: every frame becomes a keyframe)
        pool_img = torch.rand(32, 1, 3, 384, 512, generator=gen).to(device)
        mf = MotionFilter(dn, thresh=0.0, device=device)

    def feed():
        t = buf.n_frames
        if args.video_features:
            buf.fmaps[t, 0], buf.nets[t, 0], buf.inps[t, 0] = pool_f[t % 32], pool_n[t % 32], pool_i[t % 32]
        else:
            keep = mf.check(pool_img[t % 32], None)
            assert keep, "threshold 0 keeps every frame"
            buf.fmaps[t], buf.nets[t], buf.inps[t] = mf.f_fmap, mf.f_net, mf.f_inp
        if t < fe.args.warmup:  # until the frontend owns the poses: smooth trajectory along x
            buf.poses[t, 0] = 0.05 * t
            buf.disps[t, 0] = pool_d[t % 32]
        buf.n_frames += 1
        fe.run()

    fed = 0
    while not fe.is_initialized:  # warm-up: initialisation (8 keyframes, 8 update iterations), untimed
        feed()
        fed += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    u0 = fe.n_updates
    for _ in range(N - fed):
        feed()
    backend_edges = None
    if args.with_backend:  # system.py:272-275: global BA over all keyframes, twice (fresh graph each time)
        from vipe_amd.slam.backend import BackendArgs, SLAMBackend
        be = SLAMBackend(um, buf, BackendArgs(), device)
        torch.cuda.synchronize()
        t_fe = time.perf_counter() - t0
        be.run(7)
        gb = be.run(BackendArgs().backend_iters, update_depth=False)
        backend_edges = int(gb.ii.numel())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    finite = bool(torch.isfinite(buf.poses[: buf.n_frames]).all() and torch.isfinite(buf.disps[: buf.n_frames]).all())
    if rank == 0:
        print(json.dumps({
            "metric": "frames/s, keyframe frontend on a synthetic 512x384xN video (every frame a keyframe)",
            "value": world * (N - fed) / dt, "unit": "frames/s", "n_gpus": world, "steps": N - fed, "warmup": fed,
            "ms_per_step": 1e3 * dt / (N - fed), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 (correlation, GRU) + f32 geometry/BA", "data": "synthetic",
            "config": {"workload": f"{N} synthetic keyframes per clip, frontend window <= 48 edges, 4+2 update iterations "
                                   f"per keyframe, one clip per GPU", "update_iterations": fe.n_updates - u0,
                       "keyframes_kept": int(buf.n_frames), "edges_final": int(fe.graph.ii.numel()),
                       "global_ba": None if backend_edges is None else
                       {"passes": "backend.run(7) + backend.run(24), 8 GN iterations per step", "edges": backend_edges,
                        "frontend_seconds": t_fe, "backend_seconds": dt - t_fe},
                       "input": "feature maps (encoders skipped)" if args.video_features else
                                "RGB frames: motion filter + feature / context encoders in the timed region",
                       "state_finite": finite}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--keyframes", type=int, default=48)
    ap.add_argument("--extra-edges", type=int, default=0, help="seeded long-range edges on top of the radius-3 graph")
    ap.add_argument("--conv", default=os.environ.get("VIPE_AMD_CONV", "hip"), choices=["hip", "miopen"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=2)
    ap.add_argument("--mode", default="update", choices=["update", "video", "backend"],
                    help="update: the headline metric (update iterations/s on the 48-keyframe graph); video: frames/s "
                         "of the keyframe frontend on a synthetic video; backend: FactorGraph.update_batch calls/s "
                         "(hot loop B: operator over all edges + 2 GN iterations of global BA) on the same graph")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--no-hipgraph", action="store_true",
                    help="update mode: time eager launches instead of replaying the captured two-step HIP graph")
    ap.add_argument("--with-backend", action="store_true",
                    help="video mode: after the frontend pass also run the two global-BA passes of SLAMSystem.run "
                         "(backend.run(7), backend.run(24): system.py:272-275) inside the timed region")
    ap.add_argument("--video-features", action="store_true",
                    help="video mode: feed seeded feature maps instead of RGB frames (skips motion filter + encoders)")
    ap.add_argument("--also-without-gate-hoist", action="store_true",
                    help="additionally time the step with the context part of the GRU gates recomputed every iteration "
                         "(value_all_gate_work_per_iteration); off by default so that every launch of the roofline "
                         "kernel in the default command is of the same population as the event-timed ones")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist

    # one rank per GPU; the modulo only matters for rehearsing the multi-rank path on a box with fewer GPUs than ranks
    # (VIPE_BENCH_DIST_BACKEND=gloo, ranks share the card)
    device = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)

    def init_dist():
        if world > 1 and not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            backend = os.environ.get("VIPE_BENCH_DIST_BACKEND", "nccl")  # nccl = RCCL on ROCm
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)

    if args.mode == "video":
        init_dist()
        video_mode(args, device, world, rank)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.mode == "backend":
        init_dist()
        backend_mode(args, device, world, rank)
        if world > 1:
            dist.destroy_process_group()
        return

    # clip sharding: rank r owns clip r (seed differs per rank), no exchange during compute
    g, buf, graph = build_problem(device, args.keyframes, 384, 512, 3, args.extra_edges, args.conv, seed=1234 + rank)
    E = int(graph.ii.numel())

    def step():
        graph.update(t0=1, t1=args.keyframes, itrs=3)

    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            dist.barrier()

    # The update iteration has no host read-back and no shape that changes from step to step, so TWO consecutive steps
    # (the hidden state ping-pongs between two buffers) are captured once into a HIP graph and the timed region replays
    # it: the same kernels on the same stream, minus the ~45 Python / ctypes launches per step - which keeps the figure
    # independent of host jitter (eight ranks share one host's cores in the multi-GPU run; back-to-back runs on one box
    # lost up to 15 % to slow launch loops with identical kernel times).  `--no-hipgraph` times eager launches.
    launch, cg = "eager", None
    if not args.no_hipgraph and args.steps >= 2:
        try:
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
            launch = "hipgraph (2 steps per replay)"
        except Exception as e:  # noqa: BLE001 - fall back to eager launches and say so
            cg = None
            launch = f"eager (graph capture failed: {type(e).__name__})"
    # RCCL comes up only now: problem set-up, warm-up and the graph capture above are rank-local (clip sharding has no
    # data-path collective), so no communicator thread is alive while a stream is being captured
    init_dist()
    torch.cuda.synchronize()
    barrier()
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
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- the same step with ALL of the operator's work inside the iteration: by default the part of the GRU gate
    # convolutions that depends only on the per-edge context features is computed once per edge at add_factors (like
    # the correlation volume) and enters the gates as the initial accumulator value; this second figure recomputes it
    # every iteration, as the reference does
    value_no_hoist = None
    if args.also_without_gate_hoist and world == 1 and getattr(graph, "pgate", None) is not None:
        keep = graph.pgate
        graph.pgate = None
        nb = max(3, args.steps // 4)
        step()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(nb):
            step()
        torch.cuda.synchronize()
        value_no_hoist = nb / (time.perf_counter() - tb)
        graph.pgate = keep

    # ---- roofline of the dominant kernel: conv_halo32_kernel<128, 3, true> (every 3x3 convolution of the
    # flow-update operator with >= 128 output channels: corr2, z|r, q, delta0|weight0|agg1, agg2 - 5 launches per
    # step, ~50 % of the step).  Every launch of that instantiation in a few extra steps is bracketed by events on
    # the launch stream; achieved = (algorithmic flops of those launches) / (their summed duration).
    eng = graph.update_op.engine(device)
    rec = []
    achieved = gate_ms = float("nan")
    flops_per_launch = 0.0
    if eng.backend == "hip":
        orig = eng._conv

        def timed(pk, x0, x0_coff, B, H, W, *a, **k):
            cin = k.get("cin") or pk.cin
            if pk.cout > 64 and pk.kh == 3:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                orig(pk, x0, x0_coff, B, H, W, *a, **k)
                e1.record()
                rec.append((e0, e1, 2.0 * B * H * W * cin * pk.cout * pk.kh * pk.kw))
            else:
                orig(pk, x0, x0_coff, B, H, W, *a, **k)

        eng._conv = timed
        for _ in range(args.prof_steps):
            step()
        torch.cuda.synchronize()
        eng._conv = orig
        tot_ms = sum(a.elapsed_time(b) for a, b, _ in rec) or float("nan")
        tot_fl = sum(f for _, _, f in rec)
        gate_ms = tot_ms / max(1, len(rec))
        flops_per_launch = tot_fl / max(1, len(rec))
        achieved = tot_fl / (tot_ms * 1e-3) / 1e12

    # HBM traffic of the dominant kernel per launch, from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    # profiles/r01_summary.json; counters cannot be collected from inside the timed process)
    traffic = None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01_summary.json")))["void conv_halo32_kernel<128, 3, true>"]
        traffic = (prof["hbm_read_MB_per_launch"] + prof["hbm_write_MB_per_launch"]) * 1e6
    except Exception:  # noqa: BLE001
        pass

    if rank == 0:
        out = {
            "metric": "dense-BA+flow update iters/s, 512x384 48-KF graph",
            "value": world * args.steps / dt,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16 (correlation, GRU; fp32 accumulate) + f32 geometry/BA (fp64 reduced system)",
            "data": "synthetic",
            "config": {"workload": f"configs[2]-shaped: 512x384, {args.keyframes}-keyframe factor graph, E={E} edges "
                                   f"(radius-3 bidirectional), 3 GN iterations per update, one clip per GPU",
                       "conv_backend": args.conv, "parallelism": f"clip-sharded x{world}", "launch": launch,
                       "gate_context": "context-feature part of the GRU gates computed once per edge (at add_factors, "
                                       "like the correlation volume), not per iteration"
                                       if getattr(graph, "pgate", None) is not None else "recomputed every iteration"},
            "value_all_gate_work_per_iteration": value_no_hoist,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP16_TFLOPS, "traffic": traffic,
                         "kernel": "conv_halo32_kernel<128, 3, true> (NHWC fp16 implicit-GEMM 3x3 conv on MFMA 16x16x32, "
                                   "all launches with Cout >= 128 of the flow-update operator)",
                         "avg_launch_ms": gate_ms, "flops_per_launch": flops_per_launch,
                         "launches_per_step": len(rec) // max(1, args.prof_steps)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
