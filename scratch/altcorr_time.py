"""altcorr_forward alone, the four pyramid levels of the headline graph's 276 edges (fp32 maps gathered per edge, as AltCorrBlock calls it)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vipe_amd.ext import droid_net_ext
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
E, H, W, C = 276, 48, 64, 128
f1 = torch.randn(E, H, W, C, generator=g).to(dev)
for lvl in range(4):
    h, w = H >> lvl, W >> lvl
    f2 = torch.randn(E, h, w, C, generator=g).to(dev)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    coords = (torch.stack([xx, yy], -1)[None, None] + torch.randn(E, 1, H, W, 2, generator=g) * 1.5).to(dev) / 2 ** lvl
    coords = coords.contiguous()
    droid_net_ext.altcorr_forward(f1, f2, coords, 3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        droid_net_ext.altcorr_forward(f1, f2, coords, 3)
    b.record()
    torch.cuda.synchronize()
    print(f"level {lvl}: {a.elapsed_time(b) / 5 * 1e3:.0f} us")
