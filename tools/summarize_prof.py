#!/usr/bin/env python3
"""Condenses a rocprofv3 *_kernel_stats.csv into a small text table (committed under profiles/)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"# source: {sys.argv[1]}   total kernel time {tot/1e6:.2f} ms over {steps:g} steps")
print(f"{'kernel':92s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{r['Name'][:92]:92s} {r['Calls']:>6s} {int(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
