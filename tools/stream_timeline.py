"""One training step's kernels by stream from a rocprofv3 --kernel-trace CSV: wall time of the step, busy time per stream, the time during
which only ONE stream had a kernel running, and which kernels fill that time (the candidates for more overlap)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the adamw kernel
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]           # one steady-state step
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e3:.1f} us; streams: {sorted({r['Stream_Id'] for r in step})}")
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1, r)); ev.append((int(r["End_Timestamp"]), -1, r))
ev.sort(key=lambda e: (e[0], e[1]))
active = {}; last = t0; solo = collections.Counter(); conc = 0.0; idle = 0.0; solo_t = 0.0
for t, d, r in ev:
    dt = t - last
    if dt > 0:
        n = len(active)
        if n == 0: idle += dt
        elif n == 1:
            solo_t += dt
            k = next(iter(active.values()))
            solo[k["Kernel_Name"].split("(")[0][:70] + " s" + k["Stream_Id"]] += dt
        else: conc += dt
    last = t
    if d == 1: active[r["Dispatch_Id"]] = r
    else: active.pop(r["Dispatch_Id"], None)
print(f"no kernel running {idle / 1e3:.1f} us, exactly one {solo_t / 1e3:.1f} us, two or more {conc / 1e3:.1f} us")
for k, v in solo.most_common(25):
    print(f"  {v / 1e3:8.1f} us alone: {k}")
