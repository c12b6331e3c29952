#!/bin/bash
# in-step sweep of the dispatch knobs (the operands of a launch are cold in a step: a hot-loop A/B can mislead - DESIGN 5.5)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; O=gpurun_out/r05v; mkdir -p $O; rm -f $O/*.json
P="--no_cpu_baseline --no_alt_precision --no_exchange_probe --steps 20 --warmup 5"
run() { name=$1; shift; env "$@" python bench.py $P > $O/$name.json 2> $O/$name.err; }
for rep in 1 2; do
  run base_$rep TT_Q8_MIN_TILES=96
  run mt48_$rep TT_Q8_MIN_TILES=48
  run mt64_$rep TT_Q8_MIN_TILES=64
  run mt160_$rep TT_Q8_MIN_TILES=160
  run ks0_$rep TT_Q8_KSPLIT=0
  run ks2_$rep TT_Q8_KSPLIT=2
  run nohalf_$rep TT_P8_NO_HALF=1
  run tn256_$rep TT_TN_WGS=256
  run tn512_$rep TT_TN_WGS=512
done
for f in $O/*.json; do python3 - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[1].split('/')[-1]:16s} {d['ms_per_step']:.3f}")
PY
done
