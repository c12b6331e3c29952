#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05g
mkdir -p "$O"
cd "$R"
L=timetuning_amd/libtimetuning_hip.so
TT_TEST_PRINT_ERRORS=1 timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$\|amdgpu.ids\|socket.cpp\|Gloo" | tail -25 > "$O/tests.log"
cat "$O/tests.log"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_alt_precision --no_exchange_probe > "$O/c2.json" 2> "$O/c2.err"
cut -c1-1300 "$O/c2.json"
python bench.py --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 --no_cpu_baseline --no_alt_precision --no_exchange_probe > "$O/c4.json" 2> "$O/c4.err"
cut -c1-300 "$O/c4.json"
