#!/bin/bash
# round 5, second GPU call: the symmetric pair GEMM (gemm_pairs8s_kernel) against the round-4 kernel in one process, its parity tests, C2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05b
mkdir -p "$O"
cd "$R"
L=timetuning_amd/libtimetuning_hip.so
timeout 600 python tools/ab_pairs.py old=$L:TT_Q8_STREAM=0 new=$L:TT_Q8_STREAM=1 > "$O/ab_pairs.txt" 2>&1
cat "$O/ab_pairs.txt"
TT_TEST_PRINT_ERRORS=1 timeout 1500 python -m pytest tests/test_hip_pairs.py tests/test_coarse_entries.py tests/test_hip_distributed.py -q -x 2>&1 | tail -15 > "$O/tests_pairs.log"
cat "$O/tests_pairs.log"
timeout 900 python -m pytest tests/test_hip_timet.py -q -x -k "c2_full or tiny or c4_c5" 2>&1 | tail -5 > "$O/tests_timet.log"
cat "$O/tests_timet.log"
python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_alt_precision --no_exchange_probe > "$O/c2.json" 2> "$O/c2.err"
cut -c1-1500 "$O/c2.json"
