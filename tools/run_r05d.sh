#!/bin/bash
# round 5, fourth GPU call: where the symmetric kernel's time goes (ablation) + L2 / HBM counters of both persistent kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05d
mkdir -p "$O"
cd "$R"
timeout 600 python tools/q8_ablate.py stream=1 qkv proj fc1 fc2 "ideal 2 tiles/CU K384 f32out" > "$O/ablate_new.txt" 2>&1
timeout 600 python tools/q8_ablate.py stream=0 qkv fc1 >> "$O/ablate_new.txt" 2>&1
cat "$O/ablate_new.txt"
cd /tmp && export TMPDIR=/tmp
pmc() {
  local name=$1; shift
  local i=0
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $O/pmc/$name/g$i -o p -- python3 $R/tools/pairs_one.py "$@" > $O/pmc_$name.g$i.log 2>&1
  done
}
export TT_Q8_STREAM=0; pmc old_qkv gemm 25216 1152 384 pairs; pmc old_fc1 gemm 25216 1536 384 gelu
export TT_Q8_STREAM=1; pmc new_qkv gemm 25216 1152 384 pairs; pmc new_fc1 gemm 25216 1536 384 gelu; pmc new_fc2 gemm 25216 384 1536 res; pmc new_proj gemm 25216 384 384 res
cd "$R"
python3 - <<'PY'
import collections, csv, glob, os
root = "gpurun_out/r05d/pmc"
for name in sorted(os.listdir(root)):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, name, "g*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_pairs8" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    if not c: print(name, "no data"); continue
    hit = c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1)
    kc = 4.0 * c.get("SQ_WAVE_CYCLES", 1) / max(c.get("SQ_WAVES", 1), 1)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(kc, 1)
    print(f"{name:10s} L2 hit {hit:.3f} (req {c.get('TCC_REQ_sum', 0) / 1e6:.2f} M)  fetch {c.get('FETCH_SIZE', 0) * 2048 / 1e6:7.1f} MB  write {c.get('WRITE_SIZE', 0) * 1024 / 1e6:7.1f} MB  "
          f"pipe busy {busy:.3f}  parked {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}  stalled {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f}  "
          f"cycles/wave {kc:.0f}  SALU/MFMA {c.get('SQ_INSTS_SALU', 0) / max(c.get('SQ_INSTS_MFMA', 1), 1):.2f}  LDS conflict {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
PY
find "$O/pmc" -name "*.db" -delete 2>/dev/null; du -sh "$O"
