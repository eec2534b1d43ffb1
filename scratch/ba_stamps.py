"""Phase timing of ba_accum_mfma_kernel (diagnostic build, scratch/build_diag.sh). Ticks of s_memrealtime are 10 ns."""
import os, sys, ctypes
os.environ["VIPE_AMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvipe_amd_diag.so")
sys.path.insert(0, '.')
import numpy as np, torch
from vipe_amd._lib import lib
from vipe_amd.synth import make_graph
from vipe_amd.ext import slam_ext
from oracle import se3 as ose3
dev = torch.device('cuda:0')
g = make_graph()
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
E = len(g.ii); z = np.zeros_like(g.ii)
args = [T(g.poses).clone(), T(g.disps).clone(), T(g.disps_sens), T(g.intrinsics), T(ose3.se3_identity(1)),
        T(g.target.reshape(E, -1, 2)), T(g.weight.reshape(E, -1, 2)), T(g.eta), T(g.ii), T(z), T(g.jj), T(z), T(g.ii)]
L = lib()
L.vipe_diag_set_ba_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    slam_ext.dense_ba(*args, 1, 48, 1, 1e-3, 0.1)
torch.cuda.synchronize()
nblk = 12 * 48
st = torch.zeros((1 << 16) + 16, dtype=torch.int64, device=dev)
L.vipe_diag_set_ba_stamps(st.data_ptr())
slam_ext.dense_ba(*args, 1, 48, 1, 1e-3, 0.1)   # one GN iteration
torch.cuda.synchronize()
L.vipe_diag_set_ba_stamps(None)
sv = st.cpu().numpy().astype(np.int64)[(1 << 16):(1 << 16) + 6] * 0.01
print('solve phases us: load %.1f factor %.1f tail %.1f backsub %.1f retract %.1f total %.1f' % (*np.diff(sv), sv[5] - sv[0]))
s = st.cpu().numpy().astype(np.int64)[:nblk * 8].reshape(nblk, 8)
s = s[s[:, 0] > 0]
rt = s[:, :7] * 0.01
d = np.diff(rt, axis=1)
names = ["setup+tg", "walk (R1)", "finish+R2", "barrier", "per-term", "flush"]
print("blocks", len(s), "span us", rt[:, 6].max() - rt[:, 0].min())
for i, n in enumerate(names):
    print(f"  {n:12s} median {np.median(d[:, i]):7.2f} us   p90 {np.percentile(d[:, i], 90):7.2f}")
print("  total median", np.median(rt[:, 6] - rt[:, 0]))
starts = np.sort(rt[:, 0] - rt[:, 0].min())
print("  start percentiles", np.percentile(starts, [0, 50, 85, 90, 100]).round(1))
