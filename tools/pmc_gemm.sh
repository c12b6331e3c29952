#!/bin/bash
# PMC passes over the forward-GEMM shapes (development aid).  One counter group per rocprofv3 run (no trace domains).
# usage (on the GPU box, from the repo root): bash tools/pmc_gemm.sh "25216,1536,384,1"
cd /tmp && export TMPDIR=/tmp
SHAPE=${1:-25216,1536,384,1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export R
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES SQ_INST_LEVEL_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 $R/tools/gemm_shapes.py $SHAPE > $OUT/g$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob(os.environ['R'] + '/gpurun_out/pmc_gemm/g*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            if 'gemm' not in k: continue
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        for k, v in acc.items():
            print(k, {c: f"{x:.4g}" for c, x in v.items()})
PY
