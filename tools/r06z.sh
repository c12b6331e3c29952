#!/bin/bash
cd "$(dirname "$0")/.."
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GPU_MAX_HW_QUEUES=8
python -m pytest tests/test_hip_pairs.py tests/test_cpu_twin.py -q -x -m gpu 2>&1 | tail -2
python -m pytest tests/test_hip_timet.py -q -x -m gpu -k "c2_full_step or gradient_scale or c3_per_rank" 2>&1 | tail -2
S=tools/knob_sweep_step.sh
bash $S gpurun_out/r06sr/c2 TT_SPLIT_ROWS=0,1 TT_SPLIT_ROWS=0,1
BENCH_ARGS="--architecture dino-s8 --batch_size 16" bash $S gpurun_out/r06sr/c5 TT_SPLIT_ROWS=0,1
