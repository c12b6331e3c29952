#!/bin/bash
# PMC passes over the fused attention forward (development aid; one counter group per rocprofv3 run, no trace domains).
# usage (on the GPU box, from the repo root): bash tools/pmc_attn.sh "128 197 6"
cd /tmp && export TMPDIR=/tmp
ARGS=${1:-128 197 6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 $GRAFT_REPO_ROOT/tools/attn_one.py $ARGS > $OUT/g$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_attn"
for d in sorted(glob.glob(root + '/g*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:70]
            if 'attention' not in k: continue
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(k, {c: f"{sum(x)/len(x):.4g}" for c, x in v.items()})
PY
