"""Do consecutive kernel dispatches of one stream overlap?  Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and prints every
dispatch that STARTED before its predecessor (by start time, same queue) ENDED, with both kernel names."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "dispatches; queues:", sorted({r.get("Queue_Id", "?") for r in rows}), "streams:", sorted({r.get("Stream_Id", "?") for r in rows}))
prev = None
n = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None and s < prev[1]:
        n += 1
        if n <= 40:
            print(f"OVERLAP {(prev[1] - s) / 1e3:8.1f} us: {prev[2][:60]} (q{prev[3]}) still running when {r['Kernel_Name'][:60]} (q{r.get('Queue_Id','?')}) started")
    if prev is None or e > prev[1]:
        prev = (s, e, r["Kernel_Name"], r.get("Queue_Id", "?"))
print("overlapping dispatches:", n)
