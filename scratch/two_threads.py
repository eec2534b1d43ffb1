"""Two clips at once in ONE process: a thread per clip, each with its own DroidNet, SLAMSystem and stream; the global-BA
phases take turns on a lock file.  How much of the two-PROCESS gain (1.2x) does a shared address space / allocator give?"""
import os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from vipe_amd.slam.factor_graph import warm_volume_pool
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
F = 200
runners = [bench.make_clip_runner(dev) for _ in range(K)]
for r in runners:
    r(seed=10_000, n_frames=24)
torch.cuda.empty_cache(); warm_volume_pool(dev, 130)
frames = [bench.synthetic_frames(dev, k, F, 384, 512) for k in range(K)]
torch.cuda.synchronize()
res = [None] * K
lock = "/tmp/vipe_two_threads.lock"

def work(k):
    torch.cuda.set_device(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.default_stream())
    try:
        with torch.cuda.stream(s):
            res[k] = runners[k](seed=k, n_frames=F, frames=frames[k], backend_lock=lock if K > 1 else None)
            s.synchronize()
    except Exception:
        import traceback
        traceback.print_exc()

t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"K={K} threads: {K * F / dt:.1f} frames/s aggregate; per clip", [(round(r['pass1_seconds'], 2), round(r['seconds_to_global_ba_done'], 2), round(r['seconds'], 2)) for r in res], "finite", all(r['finite'] for r in res))
