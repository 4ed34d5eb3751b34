#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02
# writes raw databases under gpurun_out/<tag>_* and text summaries under gpurun_out/<tag>_summaries/ (copy those to profiles/).
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_summaries
mkdir -p $OUT
db() { find "$1" -name "*.db" | head -1; }
B="python3 bench.py --no-cpu-baseline --no-traffic"
# (a) kernel trace of the headline config, (b) one-step breakdown
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o t -- $B --steps 20 --warmup 2 > gpurun_out/${TAG}_trace.log 2>&1
python tools/rocprof_summary.py $(db gpurun_out/${TAG}_trace) $OUT/${TAG}_kernel_stats.txt > /dev/null
python tools/step_profile.py $(db gpurun_out/${TAG}_trace) > $OUT/${TAG}_step_breakdown.txt
# (c) PMC passes (separate: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2), eager launches so that every kernel is a dispatch
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${TAG}_pmc_$N -o p -- $B --steps 2 --warmup 1 --no-graph > gpurun_out/${TAG}_pmc_$N.log 2>&1
done
python tools/pmc_summary.py $(db gpurun_out/${TAG}_pmc_FETCH_SIZE) $(db gpurun_out/${TAG}_pmc_WRITE_SIZE) $(db gpurun_out/${TAG}_pmc_SQ_WAVE_CYCLES) > $OUT/${TAG}_pmc_counters.txt
# (d) configs[4] (N = 769) and configs[2]'s per-GPU share (8 complexes): kernel stats, traffic counters, bench lines
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_n769_trace -o t -- $B --residues 768 --atoms 1 --steps 6 --warmup 2 > gpurun_out/${TAG}_n769_trace.log 2>&1
python tools/rocprof_summary.py $(db gpurun_out/${TAG}_n769_trace) $OUT/${TAG}_n769_kernel_stats.txt > /dev/null
python tools/step_profile.py $(db gpurun_out/${TAG}_n769_trace) > $OUT/${TAG}_n769_step_breakdown.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${TAG}_n769_pmc_$C -o p -- $B --residues 768 --atoms 1 --steps 2 --warmup 1 --no-graph > gpurun_out/${TAG}_n769_pmc_$C.log 2>&1
done
python tools/pmc_summary.py $(db gpurun_out/${TAG}_n769_pmc_FETCH_SIZE) $(db gpurun_out/${TAG}_n769_pmc_WRITE_SIZE) > $OUT/${TAG}_n769_pmc_counters.txt
{
  $B 2>/dev/null
  $B --samples-per-gpu 8 --steps 50 --warmup 3 2>/dev/null
  $B --residues 768 --atoms 1 --steps 30 --warmup 3 2>/dev/null
  PRD_GEMM_MODE=fp32 $B 2>/dev/null
} > $OUT/${TAG}_bench_lines.jsonl
python tools/op_bench.py > $OUT/${TAG}_op_bench.txt 2>&1
python tools/train_bench.py > $OUT/${TAG}_train_bench.txt 2>&1
# (e) training step: kernel stats of three optimisation steps
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_train_trace -o t -- python3 tools/train_bench.py --steps 3 --warmup 2 > gpurun_out/${TAG}_train_trace.log 2>&1
python tools/rocprof_summary.py $(db gpurun_out/${TAG}_train_trace) $OUT/${TAG}_train_kernel_stats.txt > /dev/null
# (f) the single-track GEMM study (tools/ubench/nodegemm_bench.hip, built by hipcc beforehand) and the row-kernel phase timers
if [ -x tools/ubench/nodegemm_bench ]; then ./tools/ubench/nodegemm_bench 320 0 0 > $OUT/${TAG}_nodegemm_ubench.txt 2>&1; fi
if [ -f protein_redesign_amd/libprd_hip_timing.so ]; then
  { PRD_LIB=$PWD/protein_redesign_amd/libprd_hip_timing.so python tools/phase_timing.py tri_mul_out 320 1
    PRD_LIB=$PWD/protein_redesign_amd/libprd_hip_timing.so python tools/phase_timing.py tri_mul_proj 320 1; } > $OUT/${TAG}_row_kernel_phases.txt 2>/dev/null
fi
ls -la $OUT
# the raw rocpd databases are scratch (tens of MB): only the summaries travel back
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_pmc_* gpurun_out/${TAG}_n769_trace gpurun_out/${TAG}_n769_pmc_* gpurun_out/${TAG}_train_trace
