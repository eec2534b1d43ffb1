"""A/B of the frontend's stream knobs on the synthetic clip: operator second stream / gate overlap thresholds."""
import json, sys, time
sys.path.insert(0, ".")
import torch
import bench
from vipe_amd.slam import update_engine, factor_graph

dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 512)
orig_ue_init = update_engine.UpdateEngine.__init__
orig_fg_init = factor_graph.FactorGraph.__init__
def run(tag, side_min, gate_min):
    def ue_init(self, *a, **k):
        orig_ue_init(self, *a, **k); self.op_side_min_edges = side_min
    def fg_init(self, *a, **k):
        orig_fg_init(self, *a, **k); self.gate_overlap_min_edges = gate_min
    update_engine.UpdateEngine.__init__ = ue_init
    factor_graph.FactorGraph.__init__ = fg_init
    rc = bench.make_clip_runner(dev, height=H, width=W)
    rc(seed=10_000, n_frames=24)
    res = []
    for rep in range(3):
        r = rc(seed=rep, n_frames=200)
        res.append(r["frames"] / r["frontend_seconds"])
    print(tag, [round(x, 1) for x in res], flush=True)
for rep in range(2):
    run("side>=64 gate>=64 (default)", 64, 64)
    run("side>=16 gate>=64", 16, 64)
    run("side>=16 gate>=16", 16, 16)
    run("side>=1  gate>=64", 1, 64)
