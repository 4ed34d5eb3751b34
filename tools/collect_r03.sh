#!/bin/bash
# Round-3 evidence on the GPU box (through gpurun, from the repo root):  bash tools/collect_r03.sh
# -> gpurun_out/r03_summaries/*  (copy to profiles/)
set -u
TAG=r03
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_summaries
mkdir -p $OUT
db() { find "$1" -name "*.db" | head -1; }
B="python3 bench.py --no-cpu-baseline --no-traffic"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o t -- $B --steps 20 --warmup 2 > gpurun_out/${TAG}_trace.log 2>&1
python tools/rocprof_summary.py $(db gpurun_out/${TAG}_trace) $OUT/${TAG}_kernel_stats.txt > /dev/null
python tools/step_profile.py $(db gpurun_out/${TAG}_trace) > $OUT/${TAG}_step_breakdown.txt
PMCS=""
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${TAG}_pmc_$N -o p -- $B --steps 2 --warmup 1 --no-graph > gpurun_out/${TAG}_pmc_$N.log 2>&1
  PMCS="$PMCS $(db gpurun_out/${TAG}_pmc_$N)"
done
python tools/pmc_summary.py $PMCS > $OUT/${TAG}_pmc_counters.txt
python tools/roofline_table.py $(db gpurun_out/${TAG}_trace) $PMCS > $OUT/${TAG}_roofline.txt
cat $OUT/${TAG}_roofline.txt
# other workloads: bench lines
{
  $B 2>/dev/null
  $B --samples-per-gpu 8 --steps 50 --warmup 3 2>/dev/null
  $B --residues 768 --atoms 1 --steps 30 --warmup 3 2>/dev/null
  $B --residues 1000 --atoms 24 --steps 20 --warmup 2 2>/dev/null
  PRD_GEMM_MODE=fp32 $B 2>/dev/null
} > $OUT/${TAG}_bench_lines.jsonl
# the dominant kernel alone: both generations, counters
bash tools/ta_pmc.sh r03_v3 > /dev/null 2>&1; cp gpurun_out/r03_v3_ta_pmc.txt $OUT/${TAG}_pmc_tri_attn_core.txt
PRD_TA2_V3=0 bash tools/ta_pmc.sh r03_v2 > /dev/null 2>&1; cat gpurun_out/r03_v2_ta_pmc.txt >> $OUT/${TAG}_pmc_tri_attn_core.txt
PRD_TA_VARIANT=10 bash tools/ta_pmc.sh r03_v1 > /dev/null 2>&1; cat gpurun_out/r03_v1_ta_pmc.txt >> $OUT/${TAG}_pmc_tri_attn_core.txt
{ echo "# round-3 long-row core"; python tools/ta_long_bench.py 449 640 769 832 960 1024 2>&1 | grep N=
  echo "# first-generation long-row kernels (PRD_TA2_LONG=0)"; PRD_TA2_LONG=0 python tools/ta_long_bench.py 449 640 769 832 960 2>&1 | grep N=; } > $OUT/${TAG}_ta_long_bench.txt
# micro-benchmarks behind the kernel design
for ub in valu_rate_bench overlap_bench tile_step_bench lds_fill_bench; do
  if [ -x tools/ubench/$ub ]; then ./tools/ubench/$ub > $OUT/${TAG}_ubench_$ub.txt 2>&1; fi
done
python tools/op_bench.py > $OUT/${TAG}_op_bench.txt 2>&1
python tools/train_bench.py > $OUT/${TAG}_train_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_train_trace -o t -- python3 tools/train_bench.py --steps 3 --warmup 2 > gpurun_out/${TAG}_train_trace.log 2>&1
python tools/rocprof_summary.py $(db gpurun_out/${TAG}_train_trace) $OUT/${TAG}_train_kernel_stats.txt > /dev/null
if [ -f protein_redesign_amd/libprd_hip_timing.so ]; then
  PRD_TA2_V3=0 PRD_LIB=$PWD/protein_redesign_amd/libprd_hip_timing.so python tools/ta2_timing.py 320 > $OUT/${TAG}_tri_attn_phases.txt 2>&1
  { PRD_LIB=$PWD/protein_redesign_amd/libprd_hip_timing.so python tools/phase_timing.py tri_mul_out 320 1
    PRD_LIB=$PWD/protein_redesign_amd/libprd_hip_timing.so python tools/phase_timing.py tri_mul_proj 320 1
    PRD_LIB=$PWD/protein_redesign_amd/libprd_hip_timing.so python tools/phase_timing.py outer_linear 320 1
    PRD_LIB=$PWD/protein_redesign_amd/libprd_hip_timing.so python tools/phase_timing.py tri_mul_contract 320 1
    PRD_LIB=$PWD/protein_redesign_amd/libprd_hip_timing.so python tools/phase_timing.py pair_tail 320 1; } > $OUT/${TAG}_row_kernel_phases.txt 2>&1 || echo "phase_timing failed" >> $OUT/${TAG}_row_kernel_phases.txt
fi
ls -la $OUT
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_pmc_* gpurun_out/${TAG}_train_trace
