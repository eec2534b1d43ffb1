"""Host-side profile of the keyframe frontend (cProfile, cumulative) on the synthetic video of bench.py."""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch
import bench
dev = torch.device("cuda:0")
run_clip = bench.make_clip_runner(dev)
run_clip(seed=10_000, n_frames=24)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
r = run_clip(seed=0, n_frames=int(sys.argv[1]) if len(sys.argv) > 1 else 120)
pr.disable()
print("fps", r["frames"] / r["frontend_seconds"], "wall", time.perf_counter() - t0)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
