#!/bin/bash
cd "$(dirname "$0")/.."
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GPU_MAX_HW_QUEUES=8
python tools/attn_bwd_time.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_hip_ops.py tests/test_cpu_twin.py -q -x -m gpu -k "attention or twin" 2>&1 | tail -2
for r in 1 2 3; do python bench.py --steps 30 --warmup 10 --no_alt_precision --no_cpu_baseline --no_exchange_probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('c2', d['ms_per_step'], d['loss'])"; done
