#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM traffic per kernel.
gfx950 corrections (MI355X_MICROARCH.md, HBM): counters are in KiB-like units of 1024 B; FETCH_SIZE reports HALF the bytes
of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is taken as is.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json [git-head]]"""
import collections, csv, json, sys

def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc

f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in f:
    fb = 2.0 * 1024.0 * sum(f[k]) / len(f[k])
    wb = 1024.0 * (sum(w[k]) / len(w[k]) if k in w else 0.0)
    rows.append((fb + wb, k, len(f[k]), fb, wb))
rows.sort(reverse=True)
print(f"{'kernel':90s} {'launches':>8s} {'read MB':>9s} {'write MB':>9s}")
for tot, k, n, fb, wb in rows[:14]:
    print(f"{k[:90]:90s} {n:8d} {fb/1e6:9.1f} {wb/1e6:9.1f}")
if len(sys.argv) > 3:
    # the dominant kernel of the step (bench.py's roofline kernel, keyed by label): round 4 - gemm_pairs8_kernel (all epilogue instantiations
    # together, as the bench line books them); earlier rounds - the gemm_nt_fast instantiation with the most launches
    pairs = [r for r in rows if "gemm_pairs8" in r[1]]   # gemm_pairs8s_kernel (round 5) / gemm_pairs8_kernel
    head = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    if pairs:
        n = sum(r[2] for r in pairs)
        fb = sum(r[3] * r[2] for r in pairs) / n
        wb = sum(r[4] * r[2] for r in pairs) / n
        k, label = "gemm_pairs8s_kernel<*> (every epilogue instantiation, launch-weighted)", "gemm_pairs8s_kernel"
        sources = ["timetuning_amd/csrc/gemm_pairs8.hip", "timetuning_amd/csrc/common.hpp"]
    else:
        dom = sorted([r for r in rows if "gemm_nt_fast_kernel" in r[1]], key=lambda r: -r[2]) or rows[:1]
        tot, k, n, fb, wb = dom[0]
        names = {"<2, 2": "128x128", "<1, 2": "64x128", "<2, 1": "128x64", "<1, 1": "64x64"}   # <WM, WN[, BK]>
        label = next((f"gemm_nt_fast_kernel<{v}>" for t, v in names.items() if ("gemm_nt_fast_kernel" + t) in k), k)
        sources = ["timetuning_amd/csrc/gemm_nt_fast.hip", "timetuning_amd/csrc/common.hpp"]
    import hashlib, os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    hs = hashlib.sha256()
    for sf in sources:
        hs.update(open(os.path.join(root, sf), "rb").read())
    json.dump({"kernel": k, "kernel_label": label, "git_head": head, "kernel_sources": sources, "kernel_source_sha16": hs.hexdigest()[:16], "launches_profiled": n, "hbm_read_bytes_per_launch": round(fb), "hbm_write_bytes_per_launch": round(wb),
               "hbm_bytes_per_launch": round(fb + wb), "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 per the gfx950 correction; average over the launches of one bench.py run (mixed shapes of this kernel)"},
              open(sys.argv[3], "w"), indent=1)
