#!/bin/bash
# PMC passes over one plane-GEMM shape (development aid; one counter group per rocprofv3 run, no trace domains).
# usage (on the GPU box, from the repo root): bash tools/pmc_planes.sh "1 25216 2304 768 0 1"
cd /tmp && export TMPDIR=/tmp
ARGS=${1:-1 25216 2304 768 0 1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export R
TAG=${2:-pmc_planes}
export TAG
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 $R/tools/planes_one.py $ARGS > $OUT/g$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ["R"] + "/gpurun_out/" + os.environ["TAG"]
for d in sorted(glob.glob(root + '/g*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:70]
            if 'gemm_planes' not in k: continue
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(k, {c: f"{sum(x)/len(x):.4g}" for c, x in v.items()})
PY
