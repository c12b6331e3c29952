#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05o; mkdir -p "$O"; cd "$R"
python tools/ab_pairs.py scalar=tools/bin/libq8gelu0.so packed=timetuning_amd/libtimetuning_hip.so > "$O/ab_gelu.txt" 2>&1
python -m pytest tests/test_hip_pairs.py -m gpu -q -x 2>&1 | tail -3 > "$O/tests.log"
cat "$O/ab_gelu.txt" "$O/tests.log"
