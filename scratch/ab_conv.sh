#!/bin/bash
# alternate the library variants on ONE box: conv microbench (E=48, 276) and the headline
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for t in "$@"; do
  export VIPE_AMD_LIB=$PWD/scratch/lib/libvipe_$t.so
  python3 - <<PY 2>/dev/null
import sys; sys.path.insert(0, ".")
import torch
from vipe_amd._lib import check, lib, ptr, stream_ptr
from vipe_amd.slam.update_engine import _Packed
dev = torch.device("cuda:0"); torch.manual_seed(0)
out = []
for (H, W) in ((48, 64), (41, 73)):
  for (cin, cout) in ((320, 128), (320, 256), (128, 384)):
    w = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).half(); pk = _Packed(w, torch.zeros(cout), dev)
    for E in (48, 276):
        x = (torch.randn(E, H, W, cin) * 0.5).half().to(dev); y = torch.empty(E, H, W, cout, dtype=torch.float16, device=dev)
        run = lambda: check(lib().vipe_conv2d_nhwc_f16(ptr(x), ptr(pk.packed), ptr(pk.bias), None, ptr(y), E, H, W, cin, cin, 0, cout, cout, 0, 3, 3, 1, stream_ptr(x)), "c")
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{e0.elapsed_time(e1) / 30 * 1e3:.0f}")
print("$t conv us [48x64: 320>128 E48,E276 | 320>256 | 128>384 ; 41x73 same]:", " ".join(out))
PY
  python3 bench.py --no-secondary --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t headline', round(d['value'],1), 'it/s  roof', round(d['roofline']['frac'],3))"
done; done
