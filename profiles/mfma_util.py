"""Per-kernel matrix-core utilisation from one `rocprofv3 --pmc` pass (one row per dispatch and counter):

    python profiles/mfma_util.py <pmc_counter_collection.csv> <out.json>

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (wall cycles x 1024 SIMDs), wall cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums
the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back'); effective clock = wall cycles / dispatch duration.  The SQ wait / issue
counters are quad-cycles summed over waves: shares of SQ_WAVE_CYCLES."""
import collections
import csv
import json
import sys

src, out = sys.argv[1:3]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n.split("(")[0][:70]


per = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
calls = collections.defaultdict(set)
for r in csv.DictReader(open(src)):
    k = short(r["Kernel_Name"])
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in calls[k]:
        calls[k].add(r["Dispatch_Id"])
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {}
for k, c in per.items():
    wall = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if wall <= 0:
        continue
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    e = {"calls": len(calls[k]), "avg_us": dur[k] / len(calls[k]) / 1e3, "total_ms": dur[k] / 1e6,
         "effective_clock_GHz": wall / dur[k],
         "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (wall * 1024.0)}
    if wc > 0:
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if name in c:
                e[name.lower() + "_share"] = c[name] / wc
    if "SQ_LDS_BANK_CONFLICT" in c:
        e["lds_bank_conflict_cycles_per_call"] = c["SQ_LDS_BANK_CONFLICT"] / len(calls[k])
    res[k] = e
top = dict(sorted(res.items(), key=lambda kv: -kv[1]["total_ms"])[:16])
json.dump(top, open(out, "w"), indent=1)
print(json.dumps(top, indent=1))
