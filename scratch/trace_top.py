"""Per-kernel totals from a rocprofv3 kernel trace with readable names: python scratch/trace_top.py <kernel_trace.csv> [n] [t_from_ms] [t_to_ms]"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t0 = min(int(r["Start_Timestamp"]) for r in rows)
lo = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0
hi = float(sys.argv[4]) * 1e6 if len(sys.argv) > 4 else 1e30
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    s = int(r["Start_Timestamp"]) - t0
    if s < lo or s > hi: continue
    k = r["Kernel_Name"]
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?)E", k)
    if m: k = m.group(1)
    k = k.split("(")[0][:64]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot:.1f} ms, {sum(v[0] for v in agg.values())} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"{k:64s} {v[0]:7d} {v[1]:9.1f} ms {100*v[1]/tot:5.1f}% {1e3*v[1]/v[0]:8.1f} us")
