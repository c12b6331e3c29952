#!/bin/bash
# Copies the judged summaries of a tools/final_run.sh run (gpurun_out/final_<tag>/) into profiles/ under per-round names.
#   usage: bash tools/collect_profiles.sh <tag> [round-prefix, default r05] [git head the run was launched at]
set -euo pipefail
cd "$(dirname "$0")/.."
T=$1; R=${2:-r06}; O=gpurun_out/final_$T
for f in c2 c1 c3 c4 c4_bf16 c5; do cp "$O/$f.json" "profiles/${R}_bench_${f}_line.json"; done
sum() { python3 tools/summarize_prof.py "$1" "$2" 45 | sed "1s|\$|   ($3)|"; }
sum "$O/prof_c2/c2_kernel_stats.csv" 26 "steps incl. warm-up and the launch-by-launch profiled-GEMM pass; default precision f32-split(f16x3); three HIP streams - concurrent kernels share the chip, so a kernel's average here is NOT its isolated duration (bench.py's roofline pass times each launch alone on one stream)" > "profiles/${R}_bench_c2_kernel_stats.txt"
cp "$O/prof_c2/c2_kernel_stats.csv" "profiles/${R}_bench_c2_kernel_stats.csv"
sum "$O/prof_c2f32/c2f32_kernel_stats.csv" 26 "steps incl. warm-up and the profiled-GEMM pass; --precision f32" > "profiles/${R}_bench_c2_f32_kernel_stats.txt"
sum "$O/prof_c4/c4_kernel_stats.csv" 16 "steps incl. warm-up and the profiled-GEMM pass; f32-split(f16x3)" > "profiles/${R}_bench_c4_kernel_stats.txt"
sum "$O/prof_c5/c5_kernel_stats.csv" 11 "steps incl. warm-up and the profiled-GEMM pass; C5 per-GPU share (16 clips), f32-split(f16x3)" > "profiles/${R}_bench_c5_kernel_stats.txt"
sum "$O/prof_c4bf16/c4bf16_kernel_stats.csv" 16 "steps incl. warm-up and the profiled-GEMM pass; --precision bf16" > "profiles/${R}_bench_c4_bf16_kernel_stats.txt"
sum "$O/prof_c1/c1_kernel_stats.csv" 63 "steps incl. warm-up and the profiled-GEMM pass" > "profiles/${R}_bench_c1_kernel_stats.txt"
F=$(find "$O/pmc_fetch" -name "*counter_collection.csv" | head -1); W=$(find "$O/pmc_write" -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" profiles/dominant_kernel_traffic.json "${3:-$(git rev-parse --short HEAD)}" | head -12   # 3rd argument: the commit the run was LAUNCHED at, when HEAD has moved since
python3 tools/pmc_pairs_summary.py gpurun_out/pmc_pairs "profiles/${R}_gemm_pairs_pmc.json" "profiles/${R}_attention_pmc.json"
cp "$O/tests.log" "profiles/${R}_gpu_tests.log"
cp "$O/stream_timeline_c2.txt" "profiles/${R}_stream_timeline_c2.txt"; cp "$O/sinkhorn_device_time.txt" "profiles/${R}_sinkhorn_device_time.txt"
for v in 0 1; do cp "$O/c2_single_stream_$v.json" "profiles/${R}_bench_c2_single_stream_${v}_line.json"; done
ls profiles | grep -c "$R"
