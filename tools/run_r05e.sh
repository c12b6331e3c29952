#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05e
mkdir -p "$O"
cd "$R"
timeout 600 python tools/q8s_stamp.py > "$O/stamps.txt" 2>&1
cat "$O/stamps.txt"
