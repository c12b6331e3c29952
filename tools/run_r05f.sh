#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05f
mkdir -p "$O"
cd "$R"
L=timetuning_amd/libtimetuning_hip.so
timeout 900 python tools/ab_pairs.py l0=tools/bin/libq8s_l0.so l1=tools/bin/libq8s_l1.so l2=tools/bin/libq8s_l2.so l2p1=tools/bin/libq8s_l2p1.so > "$O/ab_pairs.txt" 2>&1
cat "$O/ab_pairs.txt"
timeout 600 python tools/q8s_stamp.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 > "$O/stamps_loader.txt"
grep "block 3 wave [07]" "$O/stamps_loader.txt"
