"""z|r hidden-state partial conv (VIPE_CONV_PARTIAL, fp32 out) vs the same conv with the plain fp16 epilogue, alone."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from vipe_amd.slam.networks import UpdateModule
dev = torch.device("cuda:0")
torch.manual_seed(0)
eng = UpdateModule().eval().engine(dev)
E, H, W = 276, 48, 64
net = torch.randn(E, H, W, 128, device=dev).tanh().half()
pg = torch.randn(E, H, W, 384, device=dev).half()
pzr = torch.empty(E, H, W, 256, dtype=torch.float32, device=dev)
y16 = torch.empty(E, H, W, 256, dtype=torch.float16, device=dev)
xbuf = torch.randn(E, H, W, 320, device=dev).relu().half()
zb = torch.empty(E, H, W, 128, dtype=torch.float16, device=dev)
rnet = torch.empty_like(zb)
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


print("partial fp32 out, accinit fp16 :", t(lambda: eng._conv(eng.zr_n, net, 0, E, H, W, mode="partial", fout=pzr, accinit=pg)))
print("plain fp16 out, accinit fp16   :", t(lambda: eng._conv(eng.zr_n, net, 0, E, H, W, y=y16, accinit=pg)))
print("plain fp16 out, no accinit     :", t(lambda: eng._conv(eng.zr_n, net, 0, E, H, W, y=y16)))
print("zr over 192 ch, accinit fp32   :", t(lambda: eng._conv(eng.zr_x, xbuf, 128, E, H, W, y=zb, y2=rnet, net=net, mode="zr", extra=extra, accinit=pzr, cin=192)))
print("zr over 320 ch, accinit fp16   :", t(lambda: eng._conv(eng.zr_s, net, 0, E, H, W, x1=xbuf, x1_coff=128, split=128, y=zb, y2=rnet, net=net, mode="zr", extra=extra, accinit=pg)))
