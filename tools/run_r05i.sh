#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05i
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
P="--no_cpu_baseline --no_alt_precision --no_exchange_probe"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c2" -o c2 -- python3 "$R/bench.py" --steps 20 --warmup 5 $P > "$O/prof_c2.log" 2>&1
cd "$R"
F=$(find "$O/prof_c2" -name "*kernel_stats.csv" | head -1); python3 tools/prof_summary.py "$F" 26 45 | cut -c1-150 | tee "$O/c2_kernel_stats.txt"
find "$O" -name "*kernel_trace.csv" -size +8M -delete
