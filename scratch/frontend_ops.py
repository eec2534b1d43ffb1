"""Which torch operators the keyframe frontend issues per frame, and from where (torch.profiler, stacks)."""
import collections, sys
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda:0")
rc = bench.make_clip_runner(dev)
rc(seed=10_000, n_frames=24)
N = 60
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    rc(seed=0, n_frames=N)
ev = prof.events()
kern = collections.Counter()
site = collections.Counter()
for e in ev:
    if e.device_type.name != "CPU" or not e.name.startswith("aten::") or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    # does this op (or its children) launch anything?
    def launches(x):
        return len(x.kernels) + sum(launches(c) for c in x.cpu_children)
    n = launches(e)
    if n == 0:
        continue
    kern[e.name] += n
    st = [s for s in (e.stack or []) if "vipe_amd" in s or "bench.py" in s]
    site[(st[0].split("/root/repo/")[-1] if st else "?")[:110] + " :: " + e.name] += n
print("launches per frame by op:")
for k, v in kern.most_common(25):
    print(f"  {v / N:6.1f}  {k}")
print("by call site:")
for k, v in site.most_common(70):
    print(f"  {v / N:6.1f}  {k}")
