"""phase stamps of ba_solve_dense_kernel (variant library built from scratch/abvar/ba_stamps.hip): cycles per phase"""
import ctypes, os, sys
sys.path.insert(0, ".")
import numpy as np, torch
from vipe_amd.ext import slam_ext
from vipe_amd.synth import make_graph
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
dll = ctypes.CDLL(os.environ["VIPE_AMD_LIB"])
for N in (16, 26):
    g = make_graph(n=N, height=96, width=128, radius=N - 1, seed=91)
    E = len(g.ii); z = np.zeros_like(g.ii)
    args = [T(g.disps_sens), T(g.intrinsics), T(np.array([[0, 0, 0, 0, 0, 0, 1.0]], np.float32)), T(g.target.reshape(E, -1, 2)),
            T(g.weight.reshape(E, -1, 2)), T(g.eta), T(g.ii), T(z), T(g.jj), T(z), T(g.ii)]
    for _ in range(10):
        poses, disps = T(g.poses).clone(), T(g.disps).clone()
        slam_ext.dense_ba(poses, disps, *args, 1, N, 2, 1e-3, 0.1)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 4096)()
    dll.vipe_dbg_dn_stamps(buf)
    s = np.array(buf[:], dtype=np.int64)
    nb = N - 1
    print(f"N={N} n={6*nb}: load {s[1]-s[0]}  loop {s[2]-s[1]}  blockinv {s[3]-s[2]}  backsub {s[4]-s[3]}  retract {s[5]-s[4]}  total {s[5]-s[0]} (ticks of s_memtime = 100 MHz? see ratio)")
    ph = np.array([[s[8 + 8*k + i + 1] - s[8 + 8*k + i] for i in range(5)] for k in range(nb)])
    print("  per-step mean wave 0: [own panel rows, barrier, next block + factor, -, barrier]:", ph.mean(0).round(0), " sum", ph.sum())
    print("  first step", ph[0], " last", ph[-1])
    for tw in range(6):
        b = 512 + 256 * tw
        wp = np.array([[s[b + 8*k + i + 1] - s[b + 8*k + i] for i in range(5)] for k in range(nb - 1)])
        print(f"  tile wave {tw} per-step mean [panel, barrier, update, extract, barrier]:", wp.mean(0).round(0), " first", wp[0], " mid", wp[nb // 2])
