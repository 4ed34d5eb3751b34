#!/bin/bash
# Round-6 evidence on the GPU box (through gpurun, from the repo root):  bash tools/collect_r06.sh
# -> gpurun_out/r05_summaries/*  (copy to profiles/)
set -u
TAG=r06
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
  rm -rf gpurun_out/${PFX}_trace
  rocprofv3 --kernel-trace --stats -d gpurun_out/${PFX}_trace -o t -- $B --steps 10 --warmup 2 "$@" > gpurun_out/${PFX}_trace.log 2>&1
  python tools/rocprof_summary.py $(db gpurun_out/${PFX}_trace) $OUT/${PFX}_kernel_stats.txt > /dev/null
  python tools/step_profile.py $(db gpurun_out/${PFX}_trace) > $OUT/${PFX}_step_breakdown.txt
  python tools/step_launches.py $(db gpurun_out/${PFX}_trace) > $OUT/${PFX}_step_launches.txt
  local PMCS=""
  for C in "${PMCSETS[@]}"; do
    local NM=$(echo $C | cut -d' ' -f1)
    rm -rf gpurun_out/${PFX}_pmc_$NM
    rocprofv3 --kernel-trace --pmc $C -d gpurun_out/${PFX}_pmc_$NM -o p -- $B --steps 2 --warmup 1 --no-graph "$@" > gpurun_out/${PFX}_pmc_$NM.log 2>&1
    PMCS="$PMCS $(db gpurun_out/${PFX}_pmc_$NM)"
  done
  python tools/pmc_summary.py $PMCS > $OUT/${PFX}_pmc_counters.txt
  echo "$PMCS"
}
# configs[1]: N = 320, one complex
P320=$(profile_shape ${TAG})
python tools/roofline_table.py $(db gpurun_out/${TAG}_trace) $P320 > $OUT/${TAG}_roofline.txt
cat $OUT/${TAG}_roofline.txt
# configs[2] per-GPU share: eight complexes per GPU
PB8=$(profile_shape ${TAG}_b8 --samples-per-gpu 8)
python tools/roofline_table.py $(db gpurun_out/${TAG}_b8_trace) $PB8 --b 8 > $OUT/${TAG}_b8_roofline.txt
cat $OUT/${TAG}_b8_roofline.txt
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
python tools/spa_bench.py > $OUT/${TAG}_spa_bench.txt 2>&1
python tools/ta_long_bench.py 449 640 769 832 961 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_ta_long_bench.txt
PRD_LIB=protein_redesign_amd/libprd_hip_timing.so python tools/ta3_timing.py 320 > $OUT/${TAG}_tri_attn_v3_phases.txt 2>&1
# training (configs[3] per-GPU share)
python tools/train_bench.py > $OUT/${TAG}_train_bench.txt 2>&1
python tools/train_bench.py --accumulate 4 --steps 3 > $OUT/${TAG}_train_bench_accumulate4.txt 2>&1
rm -rf gpurun_out/${TAG}_train_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_train_trace -o t -- python3 tools/train_bench.py --steps 3 --warmup 2 --no-launch-count > gpurun_out/${TAG}_train_trace.log 2>&1
python tools/rocprof_summary.py $(db gpurun_out/${TAG}_train_trace) $OUT/${TAG}_train_kernel_stats.txt > /dev/null
python tools/train_op_profile.py > $OUT/${TAG}_train_op_profile.txt 2>&1
python tools/train_aten_sites.py --top 60 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $OUT/${TAG}_train_aten_sites.txt
python tools/bwd_core_bench.py 2 320 > $OUT/${TAG}_bwd_core_bench.txt 2>&1
python tools/bwd_core_bench.py 1 384 >> $OUT/${TAG}_bwd_core_bench.txt 2>&1
python tools/trajectory_conditioning.py > $OUT/${TAG}_trajectory.txt 2>&1
python tools/determinism_stress.py 50 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_determinism.txt
ls -la $OUT
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_pmc_* gpurun_out/${TAG}_b8_trace gpurun_out/${TAG}_b8_pmc_* gpurun_out/${TAG}_n769_trace gpurun_out/${TAG}_n769_pmc_* gpurun_out/${TAG}_train_trace
