#!/bin/bash
# Round-4 evidence on the GPU box (through gpurun, from the repo root):  bash tools/collect_r04.sh
# -> gpurun_out/r04_summaries/*  (copy to profiles/)
set -u
TAG=r04
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_summaries
mkdir -p $OUT
db() { find "$1" -name "*.db" | head -1; }
B="python3 bench.py --no-cpu-baseline --no-traffic"
PMCSETS=("FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES")
profile_shape() {   # $1 = file prefix, $2... = bench.py shape arguments
  local PFX=$1; shift
  rocprofv3 --kernel-trace --stats -d gpurun_out/${PFX}_trace -o t -- $B --steps 10 --warmup 2 "$@" > gpurun_out/${PFX}_trace.log 2>&1
  python tools/rocprof_summary.py $(db gpurun_out/${PFX}_trace) $OUT/${PFX}_kernel_stats.txt > /dev/null
  python tools/step_profile.py $(db gpurun_out/${PFX}_trace) > $OUT/${PFX}_step_breakdown.txt
  python tools/step_launches.py $(db gpurun_out/${PFX}_trace) > $OUT/${PFX}_step_launches.txt
  local PMCS=""
  for C in "${PMCSETS[@]}"; do
    local NM=$(echo $C | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${PFX}_pmc_$NM -o p -- $B --steps 2 --warmup 1 --no-graph "$@" > gpurun_out/${PFX}_pmc_$NM.log 2>&1
    PMCS="$PMCS $(db gpurun_out/${PFX}_pmc_$NM)"
  done
  python tools/pmc_summary.py $PMCS > $OUT/${PFX}_pmc_counters.txt
  echo "$PMCS"
}
# configs[1]: N = 320
P320=$(profile_shape ${TAG})
python tools/roofline_table.py $(db gpurun_out/${TAG}_trace) $P320 > $OUT/${TAG}_roofline.txt
cat $OUT/${TAG}_roofline.txt
# configs[4]: N = 769 (the HBM evidence comes from here)
P769=$(profile_shape ${TAG}_n769 --residues 768 --atoms 1)
python tools/roofline_table.py $(db gpurun_out/${TAG}_n769_trace) $P769 --N 769 > $OUT/${TAG}_n769_roofline.txt
cat $OUT/${TAG}_n769_roofline.txt
# other workloads: bench lines
{
  $B 2>/dev/null
  $B --samples-per-gpu 8 --steps 50 --warmup 3 2>/dev/null
  $B --residues 768 --atoms 1 --steps 30 --warmup 3 2>/dev/null
  $B --residues 1000 --atoms 24 --steps 20 --warmup 2 2>/dev/null
  PRD_GEMM_MODE=fp32 $B 2>/dev/null
} > $OUT/${TAG}_bench_lines.jsonl
python bench.py > $OUT/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python tools/op_bench.py > $OUT/${TAG}_op_bench.txt 2>&1
python tools/train_bench.py > $OUT/${TAG}_train_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_train_trace -o t -- python3 tools/train_bench.py --steps 3 --warmup 2 > gpurun_out/${TAG}_train_trace.log 2>&1
python tools/rocprof_summary.py $(db gpurun_out/${TAG}_train_trace) $OUT/${TAG}_train_kernel_stats.txt > /dev/null
PRD_TRI_ATTN_BWD_V2=0 python tools/train_bench.py > $OUT/${TAG}_train_bench_fp32_bwd_core.txt 2>&1      # same-box A/B: the fp32-MFMA backward core
python tools/train_op_profile.py > $OUT/${TAG}_train_op_profile.txt 2>&1
python tools/bwd_core_bench.py 2 320 > $OUT/${TAG}_bwd_core_bench.txt 2>&1
python tools/bwd_core_bench.py 1 384 >> $OUT/${TAG}_bwd_core_bench.txt 2>&1
bash tools/bwd_pmc.sh ${TAG} > /dev/null 2>&1
cp gpurun_out/${TAG}_bwd_pmc.txt $OUT/${TAG}_bwd_core_pmc.txt
python tools/trajectory_conditioning.py > $OUT/${TAG}_trajectory.txt 2>&1
ls -la $OUT
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_pmc_* gpurun_out/${TAG}_n769_trace gpurun_out/${TAG}_n769_pmc_* gpurun_out/${TAG}_train_trace
