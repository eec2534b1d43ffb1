import sys, os, time, torch
sys.path.insert(0, '.')
from vipe_amd._lib import check, lib, ptr, stream_ptr
from vipe_amd.slam.update_engine import _Packed
dev = torch.device('cuda:0')
E, H, W = 276, 48, 64
def run(cin, cout, k, reps=10, tag=''):
    x = (torch.randn(E, H, W, cin, device=dev) * 0.5).half()
    w = (torch.randn(cout, cin, k, k) / (cin*k*k) ** 0.5).half()
    pk = _Packed(w, torch.zeros(cout), dev)
    y = torch.empty(E, H, W, cout, dtype=torch.float16, device=dev)
    def f():
        check(lib().vipe_conv2d_nhwc_f16(ptr(x), ptr(pk.packed), ptr(pk.bias), None, ptr(y), E, H, W, cin, cin, 0, cout, cout, 0, k, k, 1, stream_ptr(x)), 'conv')
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * E * H * W * cin * cout * k * k
    print(f'{tag} cin={cin} cout={cout} k={k}: {ms:.3f} ms  {fl/ms/1e9:.0f} TFLOP/s', flush=True)
for cfg in [(448, 256, 3), (448, 128, 3), (128, 128, 3), (128, 384, 3), (200, 128, 1), (128, 128, 1)]:
    run(*cfg, tag='halo')
for cfg in [(128, 64, 3), (256, 4, 3)]:
    run(*cfg, tag='small')
