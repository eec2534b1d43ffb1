"""Timing of the fused pyramid build (both layouts) and the fused lookup + 1x1 conv kernel at the bench size."""
import sys, time, torch
sys.path.insert(0, '.')
from vipe_amd.ext import droid_net_ext
from vipe_amd.slam.networks import UpdateModule
dev = torch.device('cuda:0')
E, h, w = 276, 48, 64
f1 = torch.randn(E, 128, h, w, device=dev).half(); f2 = torch.randn(E, 128, h, w, device=dev).half()
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = E * 3072 * 3072 * 2 * (1 + 1/4 + 1/16 + 1/64) / 1e9
ms = t(lambda: droid_net_ext.corr_pyramid_build(f1, f2, 4))
print(f'fused pyramid build, reference layout: {ms:.3f} ms, {gb/ms*1e3:.0f} GB/s written ({gb:.2f} GB)')
fm = torch.cat([f1[:24], f2[:24]], 0).contiguous()
i1 = torch.randint(0, 48, (E,), device=dev); i2 = torch.randint(0, 48, (E,), device=dev)
lv = droid_net_ext.corr_pyramid_build_indexed(fm, i1, i2)
ms = t(lambda: droid_net_ext.corr_pyramid_build_indexed(fm, i1, i2, levels=lv))
print(f'fused pyramid build, blocked layout (indexed): {ms:.3f} ms, {gb/ms*1e3:.0f} GB/s written')
torch.manual_seed(0)
eng = UpdateModule().eval().engine(dev)
u = torch.arange(w, device=dev).float().view(1, 1, w).expand(E, h, w); v = torch.arange(h, device=dev).float().view(1, h, 1).expand(E, h, w)
coords = (torch.stack([u, v], -1) + 3.0 * torch.randn(E, h, w, 2, device=dev)).contiguous()
out = torch.empty(E, h, w, 128, dtype=torch.float16, device=dev)
ref = droid_net_ext.corr_pyramid_build(fm[i1], fm[i2], 4)
for name, levels in (("reference", ref), ("blocked", lv)):
    ms = t(lambda: droid_net_ext.corr_lookup_conv1x1(levels, coords, eng.corr0.packed, eng.corr0.bias, out, act="relu"), 10)
    print(f'lookup + conv1x1, {name} layout: {ms*1e3:.0f} us')
