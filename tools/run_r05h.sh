#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05h
mkdir -p "$O"
cd "$R"
L=timetuning_amd/libtimetuning_hip.so
timeout 900 python tools/ab_pairs.py min128=$L:TT_Q8_MIN_TILES=128 min64=$L:TT_Q8_MIN_TILES=64 min24=$L:TT_Q8_MIN_TILES=24 > "$O/ab_pairs.txt" 2>&1
cat "$O/ab_pairs.txt"
