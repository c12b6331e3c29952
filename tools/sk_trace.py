"""Per-kernel durations and gaps of the Sinkhorn launches from a rocprofv3 --kernel-trace CSV of tools/sk_time.py."""
import csv, sys, statistics, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sk_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev = None
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void tt::", "") + f" grid{r['Grid_Size_X']}"
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append((e - s) / 1e3)
    if prev is not None and s - prev < 50000: gap[n].append((s - prev) / 1e3)
    prev = e
for n in dur:
    print(f"{n:60s} calls {len(dur[n]):5d}  median duration {statistics.median(dur[n]):6.2f} us   median gap before {statistics.median(gap[n]) if gap[n] else float('nan'):5.2f} us")
