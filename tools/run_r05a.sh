#!/bin/bash
# round 5, first GPU call: the whole GPU suite (with the error prints of the new tests) + C2 / C4-bf16 bench lines
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05a
mkdir -p "$O"
cd "$R"
TT_TEST_PRINT_ERRORS=1 timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -150 > "$O/tests.log"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_alt_precision --no_exchange_probe > "$O/c2.json" 2> "$O/c2.err"
python bench.py --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 --precision bf16 --no_cpu_baseline --no_alt_precision --no_exchange_probe > "$O/c4_bf16.json" 2> "$O/c4_bf16.err"
tail -5 "$O/tests.log"
cut -c1-600 "$O/c2.json"; echo; cut -c1-400 "$O/c4_bf16.json"
