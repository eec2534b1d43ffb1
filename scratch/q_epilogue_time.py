"""What do the fused epilogues / initial accumulators of the gate convolutions cost?  Same 320 -> 128 (q) and 320 -> 256 (z|r)
convolutions with the plain epilogue and without initial accumulators."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from vipe_amd.slam.networks import UpdateModule
dev = torch.device("cuda:0")
torch.manual_seed(0)
eng = UpdateModule().eval().engine(dev)
E, H, W = 276, 48, 64
net = torch.randn(E, H, W, 128, device=dev).tanh().half()
rnet = (torch.rand(E, H, W, 128, device=dev) * net.float()).half()
pg = torch.randn(E, H, W, 384, device=dev).half()
xbuf = torch.randn(E, H, W, 320, device=dev).relu().half()
zb = torch.rand(E, H, W, 128, device=dev).half()
r2 = torch.empty_like(zb)
nout = torch.empty_like(zb)
y256 = torch.empty(E, H, W, 256, dtype=torch.float16, device=dev)
extra = torch.zeros(E, 384, device=dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


q = dict(x1=xbuf, x1_coff=128, split=128)
print("q   fused (Q epilogue + extra + accinit):", round(t(lambda: eng._conv(eng.q_s, rnet, 0, E, H, W, y=nout, net=net, z=zb, mode="q", extra=extra, extra_off=256, accinit=pg, ai_coff=256, **q)), 4))
print("q   Q epilogue, no accinit              :", round(t(lambda: eng._conv(eng.q_s, rnet, 0, E, H, W, y=nout, net=net, z=zb, mode="q", extra=extra, extra_off=256, **q)), 4))
print("q   plain tanh epilogue + accinit       :", round(t(lambda: eng._conv(eng.q_s, rnet, 0, E, H, W, y=nout, act="tanh", accinit=pg, ai_coff=256, **q)), 4))
print("q   plain tanh epilogue, no accinit     :", round(t(lambda: eng._conv(eng.q_s, rnet, 0, E, H, W, y=nout, act="tanh", **q)), 4))
print("zr  fused (ZR epilogue + accinit)       :", round(t(lambda: eng._conv(eng.zr_s, net, 0, E, H, W, y=zb, y2=r2, net=net, mode="zr", extra=extra, accinit=pg, **q)), 4))
print("zr  plain sigmoid epilogue, no accinit  :", round(t(lambda: eng._conv(eng.zr_s, net, 0, E, H, W, y=y256, act="sigmoid", **q)), 4))
