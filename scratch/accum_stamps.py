"""phase stamps of ba_accum_mfma_kernel (one workgroup: pixel tile 3 of keyframes 2 and 24) on the headline graph, 512x384"""
import ctypes, os, sys
sys.path.insert(0, ".")
import numpy as np, torch
from vipe_amd.ext import slam_ext
from vipe_amd.synth import make_graph
dev = torch.device("cuda:0")
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
dll = ctypes.CDLL(os.environ["VIPE_AMD_LIB"])
g = make_graph(n=48, height=384, width=512, radius=3, seed=91)
E = len(g.ii); z = np.zeros_like(g.ii)
args = [T(g.disps_sens), T(g.intrinsics), T(np.array([[0, 0, 0, 0, 0, 0, 1.0]], np.float32)), T(g.target.reshape(E, -1, 2)),
        T(g.weight.reshape(E, -1, 2)), T(g.eta), T(g.ii), T(z), T(g.jj), T(z), T(g.ii)]
for _ in range(10):
    poses, disps = T(g.poses).clone(), T(g.disps).clone()
    slam_ext.dense_ba(poses, disps, *args, 1, 48, 2, 1e-3, 0.1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
dll.vipe_dbg_dn_stamps(buf)
for nm, o in (("keyframe 2", 3300), ("keyframe 24", 3316)):
    s = np.array(buf[o:o + 8], dtype=np.int64)
    d = np.diff(s)
    print(f"{nm}: setup {d[0]}  walk {d[1]}  disparity + Schur Gram {d[2]}  barrier {d[3]}  term blocks {d[4]}  frame level {d[5]}  Schur flush {d[6]}  total {s[7]-s[0]}")
