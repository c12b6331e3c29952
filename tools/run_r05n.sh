#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05n; mkdir -p "$O"; cd "$R"
python -m pytest tests -m gpu -q -x -k "attention or pairs or step_graph or sinkhorn" 2>&1 | tail -5 > "$O/tests.log"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_alt_precision > "$O/c2.json" 2> "$O/c2.err"
cat "$O/tests.log"; cut -c1-900 "$O/c2.json"
