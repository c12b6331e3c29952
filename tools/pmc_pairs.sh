#!/bin/bash
# PMC passes over the kernels of the "f16x3" mode (one counter group per rocprofv3 run, no trace domains): the persistent pair GEMM on the
# four ViT-S/16 block shapes and the pair attention kernel.  usage (GPU box, repo root): bash tools/pmc_pairs.sh ; then
# python3 tools/pmc_pairs_summary.py gpurun_out/pmc_pairs profiles/r05_gemm_pairs_pmc.json profiles/r05_attention_pmc.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_pairs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, then the program's arguments
  local name=$1; shift
  local i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
             "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "GRBM_GUI_ACTIVE GRBM_COUNT" \
             "FETCH_SIZE" \
             "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $OUT/$name/g$i -o p -- python3 $R/tools/pairs_one.py "$@" > $OUT/$name.g$i.log 2>&1
  done
}
cd $R
run qkv gemm 25216 1152 384 pairs
run proj gemm 25216 384 384 res
run fc1 gemm 25216 1536 384 gelu
run fc2 gemm 25216 384 1536 res
run attn attn 128 197 6
run attn785 attn 64 785 6
run tn tn 6304 1536 384
