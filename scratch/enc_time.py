"""Per-frame cost of the encoders / motion filter (events on the current stream)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from vipe_amd.slam.motion_filter import DroidNet, MotionFilter
dev = torch.device("cuda:0")
torch.manual_seed(0)
dn = DroidNet()
img = torch.rand(4, 1, 3, 384, 512, device=dev)
from vipe_amd.slam.encoders import normalize_images
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
x4 = normalize_images(img[0])
print("prep            %.3f ms" % timeit(lambda: normalize_images(img[0])))
print("fnet            %.3f ms" % timeit(lambda: dn.encode_features(img[0], x4)))
print("cnet            %.3f ms" % timeit(lambda: dn.encode_context(img[0], x4)))
mf = MotionFilter(dn, thresh=1e9, device=dev)
mf.check(img[0])
print("check (no kf)   %.3f ms" % timeit(lambda: mf.check(img[1])))
mf2 = MotionFilter(dn, thresh=0.0, device=dev)
mf2.check(img[0])
print("check (kf)      %.3f ms" % timeit(lambda: mf2.check(img[1])))
print("score", mf.last_score)
