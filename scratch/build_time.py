import sys, time, torch
sys.path.insert(0, '.')
from vipe_amd.ext import droid_net_ext
dev = torch.device('cuda:0')
E, h, w = 276, 48, 64
f1 = torch.randn(E, 128, h, w, device=dev).half(); f2 = torch.randn(E, 128, h, w, device=dev).half()
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: droid_net_ext.corr_pyramid_build(f1, f2, 4))
gb = E * 3072 * 3072 * 2 * (1 + 1/4 + 1/16 + 1/64) / 1e9
print(f'fused pyramid build: {ms:.3f} ms, {gb/ms*1e3:.0f} GB/s written ({gb:.2f} GB)')
def lib_path():
    vol = droid_net_ext.corr_volume(f1, f2).reshape(E * h * w, 1, h, w)
    out = [vol]
    for i in range(3):
        vol = torch.nn.functional.avg_pool2d(vol, 2, stride=2); out.append(vol)
    return out
print(f'library path (matmul + 3 avg_pool2d): {t(lib_path):.3f} ms')
