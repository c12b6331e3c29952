#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05m; mkdir -p "$O"; cd "$R"
python tools/ab_attn_pairs.py base=tools/bin/libapbase.so sp0=tools/bin/libapsp0.so ld1=tools/bin/libapld1.so ld16=tools/bin/libapld16.so ld40=tools/bin/libapld40.so 2>&1 | head -7 > "$O/ab_attn4.txt"
python tools/ap_stamp.py > "$O/stamps4.txt" 2>&1
cat "$O/ab_attn4.txt" "$O/stamps4.txt"
