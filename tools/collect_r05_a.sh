#!/bin/bash
# Round-5 evidence, part A (GPU box, from the repo root): v3 phase timing, b = 8 step profile + PMC, N = 320 step profile.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_summaries
mkdir -p $OUT
db() { find "$1" -name "*.db" | head -1; }
B="python3 bench.py --no-cpu-baseline --no-traffic"
PRD_LIB=protein_redesign_amd/libprd_hip_timing.so python tools/ta3_timing.py 320 > $OUT/r05_tri_attn_v3_phases.txt 2>&1
prof() {  # $1 prefix, rest: bench args
  local PFX=$1; shift
  rm -rf gpurun_out/${PFX}_trace
  rocprofv3 --kernel-trace --stats -d gpurun_out/${PFX}_trace -o t -- $B --steps 10 --warmup 2 "$@" > gpurun_out/${PFX}_trace.log 2>&1
  python tools/rocprof_summary.py $(db gpurun_out/${PFX}_trace) $OUT/${PFX}_kernel_stats.txt > /dev/null
  python tools/step_profile.py $(db gpurun_out/${PFX}_trace) > $OUT/${PFX}_step_breakdown.txt
  python tools/step_launches.py $(db gpurun_out/${PFX}_trace) > $OUT/${PFX}_step_launches.txt
}
prof r05
prof r05_b8 --samples-per-gpu 8
PMCSETS=("FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES")
PMCS=""
for C in "${PMCSETS[@]}"; do
  NM=$(echo $C | cut -d' ' -f1)
  rm -rf gpurun_out/r05_b8_pmc_$NM
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/r05_b8_pmc_$NM -o p -- $B --steps 2 --warmup 1 --no-graph --samples-per-gpu 8 > gpurun_out/r05_b8_pmc_$NM.log 2>&1
  PMCS="$PMCS $(db gpurun_out/r05_b8_pmc_$NM)"
done
python tools/pmc_summary.py $PMCS > $OUT/r05_b8_pmc_counters.txt
python tools/roofline_table.py $(db gpurun_out/r05_b8_trace) $PMCS --b 8 > $OUT/r05_b8_roofline.txt 2>&1
{ $B 2>/dev/null; $B --samples-per-gpu 8 --steps 50 --warmup 3 2>/dev/null; } > $OUT/r05_bench_lines_a.jsonl
cat $OUT/r05_tri_attn_v3_phases.txt; cat $OUT/r05_b8_step_breakdown.txt | head -60; cat $OUT/r05_bench_lines_a.jsonl | cut -c1-400
rm -rf gpurun_out/r05_trace gpurun_out/r05_b8_trace gpurun_out/r05_b8_pmc_*
