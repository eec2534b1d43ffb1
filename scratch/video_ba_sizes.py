"""Distribution of reduced-system shapes (free poses, block band, max degree) seen by the BA in video mode."""
import sys, collections
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from vipe_amd.ext import slam_ext
from vipe_amd.slam.buffer import GraphBuffer
from vipe_amd.slam.frontend import FrontendArgs, SLAMFrontend
from vipe_amd.slam.motion_filter import DroidNet, MotionFilter
dev = torch.device("cuda:0")
hist = collections.Counter()
orig = slam_ext.dense_ba
def patched(poses, disps, disps_sens, intrinsics, rig, target, weight, disp_damping, pi, qi, pj, qj, di, t0, t1, *a, **k):
    i, j = pi.cpu().numpy(), pj.cpu().numpy()
    band, deg = 0, 0
    for s in np.unique(i):
        grp = np.concatenate([[s], j[i == s]])
        deg = max(deg, int((i == s).sum()))
        grp = grp[(grp >= t0) & (grp < t1)]
        if grp.size:
            band = max(band, int(grp.max() - grp.min()))
    k["want_info"] = True
    info = orig(poses, disps, disps_sens, intrinsics, rig, target, weight, disp_damping, pi, qi, pj, qj, di, t0, t1, *a, **k)
    inf = info.cpu().numpy()
    hist[(int(t1 - t0), band, deg, len(i), "solver", int(inf[5]), "n", int(inf[3]))] += 1
    return info
slam_ext.dense_ba = patched
import vipe_amd.slam.buffer as B
B.slam_ext.dense_ba = patched
torch.manual_seed(0)
dn = DroidNet()
buf = GraphBuffer(384, 512, buffer_size=260, device=dev)
buf.intrinsics[:] = torch.tensor([460.8, 460.8, 256.0, 192.0], device=dev)
fe = SLAMFrontend(dn.update, buf, FrontendArgs(keyframe_thresh=0.0), dev)
gen = torch.Generator().manual_seed(99)
pool_d = (1.0 / (1.0 + 4.0 * torch.rand(32, 48, 64, generator=gen))).to(dev)
pool = torch.rand(32, 1, 3, 384, 512, generator=gen).to(dev)
mf = MotionFilter(dn, thresh=0.0, device=dev)
for f in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    t = buf.n_frames
    mf.check(pool[t % 32], None)
    buf.fmaps[t], buf.nets[t], buf.inps[t] = mf.f_fmap, mf.f_net, mf.f_inp
    if t < fe.args.warmup:
        buf.poses[t, 0] = 0.05 * t
        buf.disps[t, 0] = pool_d[t % 32]
    buf.n_frames += 1
    fe.run()
print("(n_free, block band, max degree, terms): count")
for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:25]:
    print(k, v)
