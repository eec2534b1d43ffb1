"""duration of the operator's 3x3 convolutions against the number of edges (workgroup rounds on 512 slots)"""
import sys, time
sys.path.insert(0, ".")
import torch
from vipe_amd._lib import check, lib, ptr, stream_ptr
from vipe_amd.slam.update_engine import _Packed
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (48, 64)
torch.manual_seed(0)
for (cin, cout) in ((128, 128), (320, 128), (320, 256), (128, 384)):
    w = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).half()
    pk = _Packed(w, torch.zeros(cout), dev)
    row = []
    for E in (24, 32, 40, 42, 43, 44, 48, 56, 64, 84, 85, 96, 128, 276):
        x = (torch.randn(E, H, W, cin) * 0.5).half().to(dev)
        y = torch.empty(E, H, W, cout, dtype=torch.float16, device=dev)
        def run():
            check(lib().vipe_conv2d_nhwc_f16(ptr(x), ptr(pk.packed), ptr(pk.bias), None, ptr(y), E, H, W, cin, cin, 0, cout, cout, 0, 3, 3, 1, stream_ptr(x)), "conv")
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tf = 2.0 * E * H * W * cin * cout * 9 / (us * 1e-6) / 1e12
        row.append(f"E={E}:{us:.0f}us/{tf:.0f}TF")
    print(f"{cin}->{cout}", " ".join(row), flush=True)
