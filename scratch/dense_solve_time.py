"""dense-window BA calls for a kernel-stats profile of ba_solve_dense_kernel (n = 6 (N - 1) unknowns)"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from vipe_amd.ext import slam_ext
from vipe_amd.synth import make_graph
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for N in (16, 26):
    g = make_graph(n=N, height=96, width=128, radius=N - 1, seed=91)
    E = len(g.ii); z = np.zeros_like(g.ii)
    args = [T(g.disps_sens), T(g.intrinsics), T(np.array([[0, 0, 0, 0, 0, 0, 1.0]], np.float32)), T(g.target.reshape(E, -1, 2)),
            T(g.weight.reshape(E, -1, 2)), T(g.eta), T(g.ii), T(z), T(g.jj), T(z), T(g.ii)]
    for _ in range(40):
        poses, disps = T(g.poses).clone(), T(g.disps).clone()
        slam_ext.dense_ba(poses, disps, *args, 1, N, 2, 1e-3, 0.1)
    torch.cuda.synchronize()
print("done")
