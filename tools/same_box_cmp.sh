#!/bin/bash
# Same-box comparison of this tree against an older commit's (how profiles/r05_same_box_r4_vs_r5*.txt were made).
#   1. HERE (CPU container):  bash tools/same_box_cmp.sh prepare <commit>     builds the old tree in a worktree ./_oldtree (git-excluded)
#   2. gpurun -- 'bash tools/same_box_cmp.sh run'                             alternates bench.py of both trees on ONE box
#   3. HERE:                  bash tools/same_box_cmp.sh clean
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
case "$1" in
prepare)
  git worktree add -f _oldtree "$2" -q; grep -q "^_oldtree/" .git/info/exclude 2>/dev/null || echo "_oldtree/" >> .git/info/exclude
  (cd _oldtree && python -c "import __graft_entry__ as g; g.build()") ;;
run)
  O=$R/gpurun_out/same_box; mkdir -p "$O"; P="--no_cpu_baseline --no_alt_precision --no_exchange_probe"
  one() { (cd "$1" && python bench.py $3 $P > "$O/$2.json" 2> "$O/$2.err"); }
  for rep in 1 2 3; do one $R/_oldtree c2_old_$rep "--steps 20 --warmup 5"; one $R c2_new_$rep "--steps 20 --warmup 5"; done
  C1="--steps 50 --warmup 10 --batch_size 2 --num_frames 2 --num_clusters 50"; one $R/_oldtree c1_old "$C1"; one $R c1_new "$C1"
  C3="--steps 20 --warmup 5 --use_teacher --use_queue --queue_size 2048"; one $R/_oldtree c3_old "$C3"; one $R c3_new "$C3"
  C4="--steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16"; one $R/_oldtree c4_old "$C4"; one $R c4_new "$C4"
  C5="--steps 10 --warmup 3 --architecture dino-s8 --batch_size 16"; one $R/_oldtree c5_old "$C5"; one $R c5_new "$C5"
  for f in "$O"/*.json; do python3 - "$f" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[1].split('/')[-1]:16s} {d['ms_per_step']:8.3f} ms  {d['value']:10.1f}   {r['kernel'][:28]:28s} {r['achieved']:7.2f}  {r['frac']:.4f}  {r['launches_per_step']}")
PY
  done ;;
clean) git worktree remove --force _oldtree ;;
*) echo "usage: $0 prepare <commit> | run | clean"; exit 2 ;;
esac
