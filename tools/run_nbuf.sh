#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; O=gpurun_out/r05r; mkdir -p $O; rm -f $O/*.json
P="--no_cpu_baseline --no_alt_precision --no_exchange_probe"
for rep in 1 2; do for n in 2 0; do
  TT_PAIRS_NBUF=$n python bench.py $P --steps 50 --warmup 10 --batch_size 2 --num_frames 2 --num_clusters 50 > $O/c1_n${n}_$rep.json 2> $O/c1_n${n}_$rep.err
  TT_PAIRS_NBUF=$n python bench.py $P --steps 20 --warmup 5 > $O/c2_n${n}_$rep.json 2> $O/c2_n${n}_$rep.err
done; done
for n in 2 0; do TT_PAIRS_NBUF=$n python bench.py $P --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 > $O/c4_n$n.json 2> $O/c4_n$n.err; done
for n in 2 0; do TT_PAIRS_NBUF=$n python bench.py $P --steps 10 --warmup 3 --architecture dino-s8 --batch_size 16 > $O/c5_n$n.json 2> $O/c5_n$n.err; done
for f in $O/c*.json; do python3 - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['ms_per_step'], d.get('step_graph'))
PY
done
python -m pytest tests/test_hip_pairs.py -m gpu -q -x 2>&1 | tail -2
