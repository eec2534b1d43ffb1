import sys, numpy as np, torch
sys.path.insert(0,'.')
from vipe_amd.synth import make_graph
from vipe_amd.ext import slam_ext
from oracle import se3 as ose3
dev=torch.device('cuda:0')
g=make_graph()
T=lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
E=len(g.ii); z=np.zeros_like(g.ii)
args=[T(g.poses).clone(),T(g.disps).clone(),T(g.disps_sens),T(g.intrinsics),T(ose3.se3_identity(1)),T(g.target.reshape(E,-1,2)),T(g.weight.reshape(E,-1,2)),T(g.eta),T(g.ii),T(z),T(g.jj),T(z),T(g.ii)]
r = slam_ext.dense_ba(*args, 1, 48, 1, 1e-3, 0.1)
torch.cuda.synchronize()
print('ret', r)
ws = slam_ext._ws if hasattr(slam_ext, '_ws') else None
print([k for k in dir(slam_ext) if 'ws' in k.lower() or 'work' in k.lower()])
