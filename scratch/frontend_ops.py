"""Which torch operators the keyframe frontend issues per frame, and from where (TorchDispatchMode + Python stack)."""
import collections, sys, traceback
sys.path.insert(0, ".")
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

dev = torch.device("cuda:0")
rc = bench.make_clip_runner(dev, pipelined=False)
rc(seed=10_000, n_frames=24)
N = 40
sites = collections.Counter()
ops = collections.Counter()
VIEW = ("view", "reshape", "expand", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze", "as_strided", "alias",
        "detach", "_unsafe_view", "t.", "unbind", "split", "narrow", "_local_scalar", "is_pinned", "empty", "resize", "sym_",
        "lift_fresh", "_to_copy")  # no launch (or counted through copy_)

class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        dev_arg = any(torch.is_tensor(a) and a.is_cuda for a in args) or any(
            torch.is_tensor(a) and a.is_cuda for a in (kwargs or {}).values()) or ("device" in (kwargs or {}))
        if dev_arg and not any(v in name for v in VIEW):
            st = [f for f in traceback.extract_stack(limit=14) if "/vipe_amd/" in f.filename or f.filename.endswith("bench.py")]
            site = f"{st[-1].filename.split('/')[-1]}:{st[-1].lineno} {st[-1].name}" if st else "?"
            sites[(site, name.replace("aten.", ""))] += 1
            ops[name.replace("aten.", "")] += 1
        return func(*args, **(kwargs or {}))

with Tracer():
    rc(seed=0, n_frames=N)
print("device ops per frame:", sum(ops.values()) / N)
for k, v in ops.most_common(20):
    print(f"  {v / N:6.1f}  {k}")
by_site = collections.Counter()
for (s, o), v in sites.items():
    by_site[s] += v
print("by site:")
for s, v in by_site.most_common(70):
    detail = ", ".join(f"{o}x{c / N:.1f}" for (ss, o), c in sites.items() if ss == s)
    print(f"  {v / N:6.1f}  {s}   [{detail[:150]}]")
