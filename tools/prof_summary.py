#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 --kernel-trace run (rocpd *.db, or *_kernel_stats.csv): usage prof_summary.py <db|csv> [steps] [rows]"""
import sqlite3, sys, csv
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if path.endswith(".db"):
    cur = sqlite3.connect(path).cursor()
    rows = [(n, c, t) for n, c, t in cur.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc")]
else:
    rows = [(r["Name"], int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(path))]
tot = sum(r[2] for r in rows)
print(f"# source: {path}   total kernel time {tot/1e6:.2f} ms over {steps:g} steps = {tot/1e6/steps:.3f} ms/step")
print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}")
for n, c, t in rows[:top]:
    print(f"{n[:100]:100s} {c:7d} {t/1e6:9.2f} {t/c/1e3:9.1f} {100.0*t/tot:6.2f}")
