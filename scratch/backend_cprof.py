import sys, cProfile, pstats, time
sys.path.insert(0, "/root/repo")
import torch, bench
from vipe_amd.slam.backend import BackendArgs, SLAMBackend
dev = torch.device("cuda:0")
g, buf, graph = bench.build_problem(dev, 100, 384, 512, 3, 0, "hip", seed=1234)
be = SLAMBackend(graph.update_op, buf, BackendArgs(), dev)
be.run(2)
torch.cuda.synchronize()
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
gb = be.run(7)
torch.cuda.synchronize()
pr.disable()
print("backend.run(7): %.1f ms, E=%d" % (1e3 * (time.perf_counter() - t), gb.ii.numel()))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
