"""In-kernel phase stamps (diagnostic build) of the q gate convolution: fused (Q epilogue + initial accumulators) vs plain."""
import os, sys, ctypes
os.environ["VIPE_AMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvipe_amd_diag.so")
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from vipe_amd._lib import lib
from vipe_amd.slam.networks import UpdateModule
dev = torch.device("cuda:0")
L = lib()
L.vipe_diag_set_conv_stamps.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
eng = UpdateModule().eval().engine(dev)
E, H, W = 276, 48, 64
net = torch.randn(E, H, W, 128, device=dev).tanh().half()
rnet = (torch.rand(E, H, W, 128, device=dev) * net.float()).half()
pg = torch.randn(E, H, W, 384, device=dev).half()
xbuf = torch.randn(E, H, W, 320, device=dev).relu().half()
zb = torch.rand(E, H, W, 128, device=dev).half()
nout = torch.empty_like(zb)
extra = torch.zeros(E, 384, device=dev)
q = dict(x1=xbuf, x1_coff=128, split=128)
cases = {
    "fused (Q + accinit)": lambda: eng._conv(eng.q_s, rnet, 0, E, H, W, y=nout, net=net, z=zb, mode="q", extra=extra, extra_off=256, accinit=pg, ai_coff=256, **q),
    "Q epilogue only": lambda: eng._conv(eng.q_s, rnet, 0, E, H, W, y=nout, net=net, z=zb, mode="q", extra=extra, extra_off=256, **q),
    "accinit only": lambda: eng._conv(eng.q_s, rnet, 0, E, H, W, y=nout, act="tanh", accinit=pg, ai_coff=256, **q),
    "plain": lambda: eng._conv(eng.q_s, rnet, 0, E, H, W, y=nout, act="tanh", **q),
}
nblk = E * H // 4
for name, f in cases.items():
    st = torch.zeros(nblk, 12, dtype=torch.int64, device=dev)
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    L.vipe_diag_set_conv_stamps(st.data_ptr())
    f(); torch.cuda.synchronize()
    L.vipe_diag_set_conv_stamps(None)
    s = st.cpu().numpy().astype(np.int64)
    rt = s[:, 0:10:2] * 0.01
    d = np.diff(rt, axis=1)
    print(f"{name:22s} span {rt[:, 4].max() - rt[:, 0].min():7.1f} us; per-block median us: prologue {np.median(d[:, 0]):6.2f}  kloop {np.median(d[:, 1]):6.2f}  "
          f"stage {np.median(d[:, 2]):5.2f}  epilogue {np.median(d[:, 3]):6.2f}  total {np.median(rt[:, 4] - rt[:, 0]):6.2f}", flush=True)
