#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05j
mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_hip_timet.py -q -x -k "step_graph" 2>&1 | grep -v "amdgpu.ids" | tail -30 > "$O/tests_graph.log"
cat "$O/tests_graph.log"
python bench.py --steps 50 --warmup 10 --batch_size 2 --num_frames 2 --num_clusters 50 --no_exchange_probe --no_cpu_baseline --no_alt_precision --step_graph off > "$O/c1_eager.json" 2> "$O/c1_eager.err"
cut -c1-330 "$O/c1_eager.json"; tail -3 "$O/c1_eager.err"
python bench.py --steps 50 --warmup 10 --batch_size 2 --num_frames 2 --num_clusters 50 --no_exchange_probe --no_cpu_baseline --no_alt_precision > "$O/c1.json" 2> "$O/c1.err"
cut -c1-330 "$O/c1.json"; tail -3 "$O/c1.err"
