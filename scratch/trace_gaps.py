"""GPU busy / idle accounting from a rocprofv3 kernel trace: python scratch/trace_gaps.py <kernel_trace.csv> [skip_first_ms]"""
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
t0 = ev[0][0]
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0
ev = [e for e in ev if e[0] - t0 >= skip]
busy = 0; idle = 0; gaps = []
cur_end = ev[0][0]
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, n)); idle += s - cur_end
        busy += e - s; cur_end = e
    else:
        if e > cur_end: busy += e - cur_end; cur_end = e
span = ev[-1][1] - ev[0][0]
print(f"span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), idle {idle/1e6:.1f} ms, kernels {len(ev)}")
g = np.array([x[0] for x in gaps]) / 1e3
for lo, hi in ((0, 5), (5, 20), (20, 100), (100, 1000), (1000, 1e9)):
    m = (g >= lo) & (g < hi)
    print(f"  gaps {lo:>5}-{hi:<8} us: {int(m.sum()):6d}  total {g[m].sum()/1e3:8.1f} ms")
big = collections.Counter()
for d, n in gaps:
    if d > 100e3: big[n.split("(")[0][-60:]] += d / 1e6
print("kernels that follow gaps > 100 us (ms of idle before them):")
for k, v in big.most_common(12): print(f"  {v:8.1f}  {k}")
agg = collections.Counter()
for s, e, n in ev: agg[n.split("(")[0][-70:]] += (e - s) / 1e6
print("top kernels (ms):")
for k, v in agg.most_common(14): print(f"  {v:8.1f}  {k}")
