import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
g, buf, graph = bench.build_problem(dev, 48, 384, 512, 3, 0, 'hip', seed=1234)
torch.cuda.synchronize()
for it in range(6):
    t0 = time.perf_counter()
    graph.update_batch(itrs=2, steps=1, optimize_intrinsics=False, optimize_rig_rotation=False)
    torch.cuda.synchronize()
    print('update_batch (E=%d): %.1f ms' % (graph.ii.numel(), 1e3 * (time.perf_counter() - t0)), flush=True)
