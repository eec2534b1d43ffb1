"""Which Python lines launch the small torch kernels of a frontend keyframe?  torch.profiler with stacks over a few
keyframes of the bench clip runner; prints aten ops by (count, self device time) with their innermost repo frame."""
import sys, collections
sys.path.insert(0, "/root/repo")
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
run_clip = bench.make_clip_runner(dev)
run_clip(seed=1, n_frames=30)  # warm
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    run_clip(seed=2, n_frames=40)
ka = prof.key_averages(group_by_stack_n=12)
rows2 = []
for e in ka:
    if e.key.startswith("aten::") and e.device_time_total > 0:
        fr = [f for f in e.stack if "/root/repo/" in f]
        rows2.append((e.count, e.device_time_total, e.key, fr[0].split("/root/repo/")[-1] if fr else "?"))
rows2.sort(reverse=True)
print("---- by stack")
for cnt, us, key, fr in rows2[:60]:
    print(f"{cnt / 40:6.1f}/frame {us / 40:8.1f} us/frame  {key:26s} {fr}")
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue  # count leaf ops only
    frame = "?"
    for fr in (ev.stack or []):
        if "/root/repo/vipe_amd" in fr or "/root/repo/bench.py" in fr:
            frame = fr.split("/root/repo/")[-1]
            break
    a = agg[(frame, ev.name)]
    a[0] += 1
    a[1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
print("leaf aten ops with device time:", tot, "over 40 frames =", tot / 40, "per frame")
for (frame, name), (cnt, us) in rows[:45]:
    print(f"{cnt / 40:6.1f}/frame {us / 40:8.1f} us/frame  {name:28s} {frame}")
