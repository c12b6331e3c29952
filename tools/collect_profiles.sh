#!/bin/bash
# Copies the judged summaries of a tools/final_run.sh run (gpurun_out/final_<tag>/) into profiles/ under round-3 names.
#   usage: bash tools/collect_profiles.sh <tag> [round-prefix, default r03]
set -euo pipefail
cd "$(dirname "$0")/.."
T=$1; R=${2:-r03}; O=gpurun_out/final_$T
for f in c2 c1 c3 c4 c4_f32 c5 c2_bf16x6; do cp "$O/$f.json" "profiles/${R}_bench_${f}_line.json"; done
steps() { python3 - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(d["steps"] + d["warmup"] + 1)
PY
}
sum() { python3 tools/summarize_prof.py "$1" "$2" 45 | sed "1s|\$|   ($3)|"; }
sum "$O/prof_c2/c2_kernel_stats.csv" 26 "steps incl. warm-up and the profiled-GEMM pass" > "profiles/${R}_bench_c2_kernel_stats.txt"
cp "$O/prof_c2/c2_kernel_stats.csv" "profiles/${R}_bench_c2_kernel_stats.csv"
sum "$O/prof_c4/c4_kernel_stats.csv" 14 "steps incl. warm-up and the profiled-GEMM pass" > "profiles/${R}_bench_c4_bf16_kernel_stats.txt"
sum "$O/prof_c2x6/c2x6_kernel_stats.csv" 26 "steps incl. warm-up and the profiled-GEMM pass" > "profiles/${R}_bench_c2_bf16x6_kernel_stats.txt"
sum "$O/prof_c1/c1_kernel_stats.csv" 61 "steps incl. warm-up and the profiled-GEMM pass" > "profiles/${R}_bench_c1_kernel_stats.txt"
F=$(find "$O/pmc_fetch" -name "*counter_collection.csv" | head -1); W=$(find "$O/pmc_write" -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" profiles/dominant_kernel_traffic.json "$(git rev-parse --short HEAD)" | head -12
cp "$O/tests.log" "profiles/${R}_gpu_tests.log"
ls -la profiles | grep "$R" | wc -l
