"""hipGraph capture of two consecutive update iterations (the hidden state ping-pongs between two buffers)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda:0")
g, buf, graph = bench.build_problem(dev, 48, 384, 512, 3, 0, "hip", seed=1234)
def step():
    graph.update(t0=1, t1=48, itrs=3)
for _ in range(4):
    step()
torch.cuda.synchronize()
# eager timing
t = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("eager   : %.3f ms/step" % ((time.perf_counter() - t) / 20 * 1e3))
# capture
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step(); step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
p0, d0 = buf.poses[:48].clone(), buf.disps[:48].clone()
cg = torch.cuda.CUDAGraph()
with torch.cuda.graph(cg):
    step(); step()
torch.cuda.synchronize()
print("captured")
cg.replay(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): cg.replay()
torch.cuda.synchronize()
print("graph   : %.3f ms/step" % ((time.perf_counter() - t) / 20 * 1e3))
print("finite:", bool(torch.isfinite(buf.poses[:48]).all()), "moved:", float((buf.poses[:48] - p0).abs().max()))
