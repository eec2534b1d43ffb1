"""headline it/s (graph replay) against the share of edges whose z|r hidden-state part is staged under the BA and the
per-iteration piece fractions (round 4: the delay kernel is gone - does the optimum move?)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
dev = torch.device("cuda", 0)
g, buf, graph = bench.build_problem(dev, 48, 384, 512, 3, 0)
step = lambda: graph.update(t0=1, t1=48, itrs=3)
for share, fr in ((0.5, [0.4, 0.4, 0.2]), (0.4, [0.4, 0.4, 0.2]), (0.6, [0.4, 0.4, 0.2]), (0.7, [0.4, 0.4, 0.2]), (0.5, [0.34, 0.33, 0.33]),
                  (0.5, [0.5, 0.4, 0.1]), (0.6, [0.45, 0.4, 0.15]), (0.5, [0.4, 0.4, 0.2])):
    graph.gate_overlap_share, graph.gate_overlap_fractions = share, fr
    graph._gate_state = None
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    cg = bench.capture_two_steps(step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        cg.replay()
    torch.cuda.synchronize()
    print(f"share {share} fractions {fr}: {40 / (time.perf_counter() - t0):.1f} it/s")
    del cg
