"""What the loads of each pyramid level cost the fused lookup (the upper bound of what computing a level on the fly could
return): the kernel timed alone on the headline graph, per library variant (VIPE_AMD_LIB = a -DVIPE_LOOKUP_SKIP_LEVEL=l build)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
dev = torch.device("cuda")
g, buf, graph = bench.build_problem(dev, 48, 384, 512, 3, 0)
step = lambda: graph.update(t0=1, t1=48, itrs=3)
for _ in range(3):
    step()
r = bench.hbm_kernel_rooflines(graph, step, dev, n=10)
k = r["corr_lookup_conv_kernel"]
print(os.environ.get("VIPE_AMD_LIB", "base").split("_")[-1], "lookup ms", round(k["avg_launch_ms"], 4), "frac", round(k["frac"], 3),
      "| pyramid build ms", round(r["corr_pyramid_build_kernel"]["avg_launch_ms"], 3), "| accum ms", round(r["ba_accum_mfma_kernel"]["avg_launch_ms"], 4))
