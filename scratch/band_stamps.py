"""phase stamps of the two-chain band solve (variant library from scratch/make_ba_stamps.py) on the headline graph"""
import ctypes, os, sys
sys.path.insert(0, ".")
import numpy as np, torch
from vipe_amd.ext import slam_ext
from vipe_amd.synth import make_graph
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
dll = ctypes.CDLL(os.environ["VIPE_AMD_LIB"])
for N, radius in ((48, 3), (48, 2), (32, 3)):
    g = make_graph(n=N, height=96, width=128, radius=radius, seed=91)
    E = len(g.ii); z = np.zeros_like(g.ii)
    args = [T(g.disps_sens), T(g.intrinsics), T(np.array([[0, 0, 0, 0, 0, 0, 1.0]], np.float32)), T(g.target.reshape(E, -1, 2)),
            T(g.weight.reshape(E, -1, 2)), T(g.eta), T(g.ii), T(z), T(g.jj), T(z), T(g.ii)]
    for _ in range(10):
        poses, disps = T(g.poses).clone(), T(g.disps).clone()
        slam_ext.dense_ba(poses, disps, *args, 1, N, 2, 1e-3, 0.1)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 4096)()
    dll.vipe_dbg_dn_stamps(buf)
    s = np.array(buf[3000:3008], dtype=np.int64)
    d = np.diff(s)
    for nm, o in (("start", 3100), ("rows done", 3120), ("at barrier", 3140)):
        print("   per wave", nm, (np.array(buf[o:o+16], dtype=np.int64) - s[0]).tolist())
    e = np.array(buf[3008:3011], dtype=np.int64)
    print(f"   load detail: matrix rows {e[0]-s[0]}  rhs {e[1]-e[0]}  descriptors {e[2]-e[1]}  barrier {s[1]-e[2]}")
    print(f"N={N} radius={radius} E={E}: load {d[0]}  factor0 {d[1]}  chains {d[2]}  merge {d[3]}+separator  sep-backsub {d[4]}  chain-backsub {d[5]}  out+retract {d[6]}  total {s[7]-s[0]}")
