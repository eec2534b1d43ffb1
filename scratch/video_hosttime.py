"""How much of a keyframe's wall time does the host need?  Process CPU time (user + system) of the timed clip against its
wall time; and the time the host spends blocked in the three stream synchronisations of a keyframe."""
import sys, time, os
sys.path.insert(0, "/root/repo")
import torch
import bench
dev = torch.device("cuda:0")
run_clip = bench.make_clip_runner(dev)
run_clip(seed=10_000, n_frames=24)
torch.cuda.synchronize()
waits = [0.0]
orig_sync = torch.cuda.Event.synchronize
def timed_sync(self):
    t = time.perf_counter(); orig_sync(self); waits[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = timed_sync
c0 = time.process_time(); t0 = time.perf_counter()
r = run_clip(seed=0, n_frames=200)
c1 = time.process_time(); t1 = time.perf_counter()
print(f"frames/s {r['frames'] / r['frontend_seconds']:.1f}  wall {t1 - t0:.3f} s  process CPU {c1 - c0:.3f} s  "
      f"blocked in Event.synchronize {waits[0]:.3f} s")
