"""Timeline of the last full update iteration in a rocprofv3 --kernel-trace database (rocpd sqlite): start / end / duration
per kernel with the queue it ran on.  usage: trace_step.py <results.db> [n_kernels_back] [n_kernels]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 75
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 50
sel = rows[len(rows) - back:len(rows) - back + cnt]
t0 = sel[0][1]
for name, st, en, q in sel:
    nm = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", name)[:44]
    print(f"{(st - t0) / 1e3:9.1f} {(en - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{q} {nm}")
