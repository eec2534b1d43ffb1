"""Where does the pyramid build spend its time?  Levels 1..4 written, both layouts, 276 edges of the 48 x 64 grid."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from vipe_amd.ext import droid_net_ext as D
dev = torch.device("cuda:0")
torch.manual_seed(0)
fm = torch.randn(48, 128, 48, 64, device=dev).half()
E = 276
i1 = torch.randint(0, 48, (E,), device=dev)
i2 = torch.randint(0, 48, (E,), device=dev)
for layout in (D.BLOCKED, 0):
    for nl in (1, 2, 3, 4):
        lv = D.corr_pyramid_build_indexed(fm, i1, i2, layout=layout, num_levels=nl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            D.corr_pyramid_build_indexed(fm, i1, i2, levels=lv, layout=layout, num_levels=nl)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        nbytes = sum(x.numel() * 2 for x in lv)
        print(f"layout {layout} levels {nl}: {dt * 1e3:.3f} ms  {nbytes / dt / 1e12:.2f} TB/s", flush=True)
        del lv
