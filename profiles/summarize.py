"""Turn rocprofv3 outputs (kernel stats CSV + PMC counter CSVs) into the per-kernel summary committed next to them.

    python profiles/summarize.py <kernel_stats.csv> <fetch_counter.csv> <write_counter.csv> <out.json> [steps]

FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md section HBM);
both counters are in KiB."""
import collections
import csv
import json
import sys

stats, fetch, write, out = sys.argv[1:5]
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 30  # bench.py default: 3 warm-up + 20 timed + 3 + 4 event-timed steps


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n.split("(")[0][:70]


res = collections.OrderedDict()
for r in csv.DictReader(open(stats)):
    k = short(r["Name"])
    res[k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
              "pct": float(r["Percentage"]), "ms_per_step": float(r["TotalDurationNs"]) / 1e6 / steps}
for path, key, mult in ((fetch, "hbm_read_MB_per_launch", 2.0), (write, "hbm_write_MB_per_launch", 1.0)):
    agg = collections.defaultdict(list)
    try:
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    except FileNotFoundError:
        continue
    for k, v in agg.items():
        if k in res:
            res[k][key] = mult * sum(v) / len(v) * 1024 / 1e6
json.dump({k: v for k, v in list(res.items())[:16]}, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in list(res.items())[:8]}, indent=1))
