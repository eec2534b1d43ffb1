"""phase stamps of ba_walk_kernel (pixel tile 3 of every source frame, last launch) after a short synthetic video run"""
import ctypes, os, subprocess, sys
sys.path.insert(0, ".")
import numpy as np, torch
dll = ctypes.CDLL(os.environ["VIPE_AMD_LIB"])
import bench
sys.argv = ["bench.py", "--mode", "video", "--frames", "60"]
try:
    bench.main()
except SystemExit:
    pass
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
dll.vipe_dbg_dn_stamps(buf)
for y in range(16):
    s = np.array(buf[3400 + 32 * y: 3400 + 32 * y + 32], dtype=np.int64)
    if s[0] == 0: continue
    deg = int(s[31]); nch = (deg + 11) // 12
    out = [f"frame slot {y}: terms {deg}"]
    t0 = s[0]
    for c in range(nch):
        b = s[6 * c + 1: 6 * c + 6]
        prev = s[0] if c == 0 else s[6 * (c - 1) + 5]
        out.append(f"chunk {c}: wait {b[0]-prev} setup {b[1]-b[0]} terms {b[2]-b[1]} barrier {b[3]-b[2]} flush {b[4]-b[3]}")
    out.append(f"tail barrier {s[20]-s[6*(nch-1)+5]} frame level + disparity {s[21]-s[20]} total {s[21]-s[0]}")
    print("  ".join(out))
