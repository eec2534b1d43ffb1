import sys, time, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
run_clip = bench.make_clip_runner(dev)
run_clip(seed=10_000, n_frames=24)
for i in range(3):
    r = run_clip(seed=i, n_frames=200, with_backend=True)
    print(i, "frontend", round(r["frontend_seconds"], 3), "backend", round(r["seconds"] - r["frontend_seconds"], 3), "edges", r["backend_edges"],
          "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 1))
