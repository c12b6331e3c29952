#!/bin/bash
# round 6: label-propagation similarities on the side stream (A/B), small grids on the four-wave kernel (A/B), the tests of both
cd "$(dirname "$0")/.."
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06m; mkdir -p $O
python -m pytest tests/test_hip_ops.py -q -x -m gpu -k "label_prop" > $O/tests_lp.log 2>&1; tail -2 $O/tests_lp.log
python -m pytest tests/test_hip_timet.py -q -x -m gpu -k "two_streams or step_graph_equals" > $O/tests_streams.log 2>&1; tail -2 $O/tests_streams.log
L=timetuning_amd/libtimetuning_hip.so
TT_AB_CASES=small python tools/ab_pairs.py general=$L:TT_Q4_SMALL=0 q4whole=$L:TT_Q4_SMALL=1 q4half=$L:TT_Q4_SMALL=2 > $O/ab_small.txt 2>&1; cat $O/ab_small.txt
for r in 1 2 3; do
  for v in 1 0; do
    TT_LP_SIMS_SIDE=$v python bench.py --steps 30 --warmup 10 --no_alt_precision --no_cpu_baseline --no_exchange_probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('sims_side=$v', d['ms_per_step'], d['loss'])"
  done
done 2>&1 | tee $O/ab_sims.txt
for r in 1 2; do
  for v in 0 1 2; do
    TT_Q4_SMALL=$v python bench.py --steps 30 --warmup 10 --no_alt_precision --no_cpu_baseline --no_exchange_probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('q4_small=$v', d['ms_per_step'], d['loss'])"
  done
done 2>&1 | tee $O/ab_q4small.txt
