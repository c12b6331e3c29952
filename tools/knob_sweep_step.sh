#!/bin/bash
# In-step sweep of dispatch knobs that were tuned launch by launch on one stream: with the step's chains on three streams the best setting
# may differ (a weight-gradient launch now shares the chip with the data-gradient chain).  bench.py C2, alternating, ms/step + loss.
#   usage: bash tools/knob_sweep_step.sh <out dir> KNOB=v1,v2,... [KNOB=...]    (the first value of each list is the default's stand-in)
cd "$(dirname "$0")/.."
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GPU_MAX_HW_QUEUES=8
O=$1; shift; mkdir -p $O
one() { env "$@" python bench.py --steps 30 --warmup 10 --no_alt_precision --no_cpu_baseline --no_exchange_probe ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', d['ms_per_step'], d['loss'])"; }
for spec in "$@"; do
  k=${spec%%=*}; vs=${spec#*=}
  for r in 1 2; do
    for v in ${vs//,/ }; do one $k=$v; done
  done
done 2>&1 | tee $O/sweep.txt
