#!/bin/bash
# round 5, third GPU call: cache policy of the epilogue stores (sc1 / nt / plain) of gemm_pairs8s_kernel - timing A/B + L2 / HBM counters
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05c
mkdir -p "$O"
cd "$R"
L=timetuning_amd/libtimetuning_hip.so
timeout 900 python tools/ab_pairs.py old=$L:TT_Q8_STREAM=0 sc1=$L:TT_Q8_STREAM=1 plain=tools/bin/libq8s_plain.so nt=tools/bin/libq8s_nt.so > "$O/ab_pairs.txt" 2>&1
cat "$O/ab_pairs.txt"
timeout 600 python -m pytest tests/test_hip_distributed.py -q -x -k "gloo-4 or gloo-2" 2>&1 | grep -v "^$" | tail -60 > "$O/tests_dist.log"
tail -30 "$O/tests_dist.log"
cd /tmp && export TMPDIR=/tmp
pmc() {  # variant-name, env assignments..., -- shape args
  local name=$1; shift
  local i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $O/pmc/$name/g$i -o p -- python3 $R/tools/pairs_one.py "$@" > $O/pmc_$name.g$i.log 2>&1
  done
}
export TT_Q8_STREAM=0; pmc old_qkv gemm 25216 1152 384 pairs; pmc old_fc1 gemm 25216 1536 384 gelu
export TT_Q8_STREAM=1; pmc sc1_qkv gemm 25216 1152 384 pairs; pmc sc1_fc1 gemm 25216 1536 384 gelu; pmc sc1_fc2 gemm 25216 384 1536 res
export TT_LIB_PATH=$R/tools/bin/libq8s_plain.so; pmc plain_qkv gemm 25216 1152 384 pairs; pmc plain_fc1 gemm 25216 1536 384 gelu
unset TT_LIB_PATH
cd "$R"
python3 - <<'PY'
import collections, csv, glob, os
root = "gpurun_out/r05c/pmc"
for name in sorted(os.listdir(root)):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, name, "g*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_pairs8" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    if not c: print(name, "no data"); continue
    hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(4.0 * c.get("SQ_WAVE_CYCLES", 1) / max(c.get("SQ_WAVES", 1), 1), 1)
    print(f"{name:12s} L2 hit {hit:.3f}  fetch {c.get('FETCH_SIZE', 0) * 2048 / 1e6:7.1f} MB  write {c.get('WRITE_SIZE', 0) * 1024 / 1e6:7.1f} MB  "
          f"matrix pipe busy {busy:.3f}  parked {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
find "$O/pmc" -name "*.db" -delete 2>/dev/null; du -sh "$O"
