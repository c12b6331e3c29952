#!/bin/bash
# The round's evidence in ONE gpurun call: GPU tests, the bench lines of C1..C5, rocprofv3 kernel tables and the PMC passes.
#   usage (from the repo root on the GPU box):  bash tools/final_run.sh [tag]      (tag names the outputs, default r05)
# Every profiled program follows `--` directly as python3 (no env / bash hop: the profiler initialises the GPU before the program starts).
set -uo pipefail
# what `import timetuning_amd` sets before the first HIP call - exported here because under rocprofv3 the profiler's library may initialise the
# runtime before the program starts (timetuning_amd/__init__.py says why each is needed)
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GPU_MAX_HW_QUEUES=8
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-r06}
O=$R/gpurun_out/final_$T
mkdir -p "$O"
cd "$R"
python -m pytest tests -m gpu -q 2>&1 | tail -8 > "$O/tests.log"
B="python bench.py"
$B --steps 20 --warmup 5 > "$O/c2.json" 2> "$O/c2.err"                                                   # headline: f32-split(f16x3), alt: f32 / bf16x6 / bf16
$B --steps 50 --warmup 10 --batch_size 2 --num_frames 2 --num_clusters 50 --no_exchange_probe > "$O/c1.json" 2> "$O/c1.err"
$B --steps 20 --warmup 5 --use_teacher --use_queue --queue_size 2048 --no_cpu_baseline --no_alt_precision > "$O/c3.json" 2> "$O/c3.err"
$B --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 --no_exchange_probe > "$O/c4.json" 2> "$O/c4.err"
$B --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 --precision bf16 --no_alt_precision --no_exchange_probe > "$O/c4_bf16.json" 2> "$O/c4_bf16.err"
$B --steps 10 --warmup 3 --architecture dino-s8 --batch_size 16 --no_cpu_baseline --no_exchange_probe > "$O/c5.json" 2> "$O/c5.err"
cd /tmp && export TMPDIR=/tmp
P="--no_cpu_baseline --no_alt_precision --no_exchange_probe"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c2" -o c2 -- python3 "$R/bench.py" --steps 20 --warmup 5 $P > "$O/prof_c2.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c2f32" -o c2f32 -- python3 "$R/bench.py" --steps 20 --warmup 5 --precision f32 $P > "$O/prof_c2f32.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c4" -o c4 -- python3 "$R/bench.py" --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 $P > "$O/prof_c4.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c5" -o c5 -- python3 "$R/bench.py" --steps 6 --warmup 2 --architecture dino-s8 --batch_size 16 $P > "$O/prof_c5.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c4bf16" -o c4bf16 -- python3 "$R/bench.py" --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 --precision bf16 $P > "$O/prof_c4bf16.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c1" -o c1 -- python3 "$R/bench.py" --steps 50 --warmup 10 --batch_size 2 --num_frames 2 --num_clusters 50 $P > "$O/prof_c1.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -o f -- python3 "$R/bench.py" --steps 3 --warmup 1 $P > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -o w -- python3 "$R/bench.py" --steps 3 --warmup 1 $P > "$O/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$O/trace_c2" -o t -- python3 "$R/bench.py" --steps 6 --warmup 2 $P > "$O/trace_c2.log" 2>&1
cd "$R"
python3 tools/stream_timeline.py "$O/trace_c2/t_kernel_trace.csv" > "$O/stream_timeline_c2.txt" 2>&1; rm -f "$O/trace_c2/t_kernel_trace.csv"
python3 tools/sk_time.py 6272 200 10 > "$O/sinkhorn_device_time.txt" 2>&1
for v in 0 1; do TT_SINGLE_STREAM=$v $B --steps 20 --warmup 5 $P > "$O/c2_single_stream_$v.json" 2>> "$O/c2.err"; done   # the streams' A/B on this box
bash tools/pmc_pairs.sh > "$O/pmc_pairs.log" 2>&1
find "$O" "$R/gpurun_out/pmc_pairs" -name "*kernel_trace.csv" -size +8M -delete
find "$R/gpurun_out/pmc_pairs" -name "*.db" -delete 2>/dev/null
du -sh "$O" "$R/gpurun_out/pmc_pairs"
cat "$O/tests.log"
for f in c2 c1 c3 c4 c4_bf16 c5; do echo "== $f"; cut -c1-700 "$O/$f.json"; echo; done
