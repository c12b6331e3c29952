#!/bin/bash
# round 5: persistent pair attention - tests, A/B against the round-4 kernel, and the changed step-graph / Sinkhorn tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05l; mkdir -p "$O"; cd "$R"
python -m pytest tests -m gpu -q -x -k "attention or step_graph or sinkhorn" 2>&1 | tail -5 > "$O/tests.log"
python tools/ab_attn_pairs.py base=tools/bin/libapbase.so p1=timetuning_amd/libtimetuning_hip.so:TT_ATTN_PAIRS_PERSIST=1 p0=timetuning_amd/libtimetuning_hip.so:TT_ATTN_PAIRS_PERSIST=0 > "$O/ab_attn.txt" 2>&1
cat "$O/tests.log" "$O/ab_attn.txt"
