#!/bin/bash
# PMC counters of the triangle attention backward cores (tools/bwd_core_bench.py, b = 2, N = 320):
#   bash tools/bwd_pmc.sh <tag>        -> gpurun_out/<tag>_bwd_pmc.txt
set -u
TAG=${1:-bwd}
export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
OUT=gpurun_out/${TAG}_bwd_pmc.txt
: > $OUT
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  rm -rf gpurun_out/${TAG}_pmc_$N
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${TAG}_pmc_$N -o p -- python3 tools/bwd_core_bench.py 2 320 > gpurun_out/${TAG}_pmc_$N.log 2>&1
  python tools/pmc_summary.py $(db gpurun_out/${TAG}_pmc_$N) | grep -i "kernel \|tri_attn_bwd" >> $OUT
  rm -rf gpurun_out/${TAG}_pmc_$N
done
cat $OUT
