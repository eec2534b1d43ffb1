"""In-kernel phase timing of the halo conv (diagnostic build, scratch/build_diag.sh).
Phases: 0 entry, 1 prologue done, 2 K loop done, 3 staged, 4 end.  s_memrealtime ticks are 10 ns."""
import os, sys, ctypes
os.environ["VIPE_AMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvipe_amd_diag.so")
sys.path.insert(0, '.')
import numpy as np, torch
from vipe_amd._lib import check, lib, ptr, stream_ptr
from vipe_amd.slam.update_engine import _Packed
dev = torch.device('cuda:0')
E, H, W = 276, 48, 64
L = lib()
L.vipe_diag_set_conv_stamps.argtypes = [ctypes.c_void_p]
def run(cin, cout, k):
    x = (torch.randn(E, H, W, cin, device=dev) * 0.5).half()
    w = (torch.randn(cout, cin, k, k) / (cin*k*k) ** 0.5).half()
    pk = _Packed(w, torch.zeros(cout), dev)
    y = torch.empty(E, H, W, cout, dtype=torch.float16, device=dev)
    nblk = E * H // 4 * ((cout + 127) // 128)
    st = torch.zeros(nblk, 12, dtype=torch.int64, device=dev)
    def f():
        check(L.vipe_conv2d_nhwc_f16(ptr(x), ptr(pk.packed), ptr(pk.bias), None, ptr(y), E, H, W, cin, cin, 0, cout, cout, 0, k, k, 1, stream_ptr(x)), 'conv')
    for _ in range(20): f()
    torch.cuda.synchronize()
    L.vipe_diag_set_conv_stamps(st.data_ptr())
    f(); torch.cuda.synchronize()
    L.vipe_diag_set_conv_stamps(None)
    s = st.cpu().numpy().astype(np.int64)
    rt = s[:, 0:10:2] * 0.01  # us
    cy = s[:, 1:10:2]
    t0 = rt[:, 0].min()
    d = np.diff(rt, axis=1)
    clk = (cy[:, 2] - cy[:, 1]) / np.maximum(d[:, 1], 1e-9) / 1e3  # GHz over K loop
    print(f"cin={cin} cout={cout} k={k}: blocks={nblk} span={rt[:,4].max()-t0:.1f} us; per-block median us: prologue {np.median(d[:,0]):.2f}  kloop {np.median(d[:,1]):.2f}  stage {np.median(d[:,2]):.2f}  epilogue {np.median(d[:,3]):.2f}  total {np.median(rt[:,4]-rt[:,0]):.2f}; clock {np.median(clk):.2f} GHz", flush=True)
    kc = cy[:, 2] - cy[:, 1]
    if k == 3:
        print(f"   wave 0 inside the K loop (share of its cycles): waiting for LDS-DMA (s_waitcnt vmcnt) {np.median(s[:,10]/np.maximum(kc,1)):.3f}  "
              f"at the per-tap barrier {np.median(s[:,11]/np.maximum(kc,1)):.3f}")
    # gaps between consecutive blocks on a CU are not visible here; report the first-wave start distribution
    starts = np.sort(rt[:, 0] - t0)
    print("   start time percentiles (us):", np.percentile(starts, [0, 5, 25, 50, 75, 100]).round(1))
for cfg in [(448, 256, 3), (448, 128, 3), (128, 128, 3), (128, 384, 3), (200, 128, 1), (128, 64, 3)]:
    run(*cfg)
