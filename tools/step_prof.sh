#!/bin/bash
# One replayed step under rocprofv3 (GPU box, from the repo root):  bash tools/step_prof.sh <tag> [bench.py args]
#   -> gpurun_out/<tag>_kernel_stats.txt, <tag>_step_breakdown.txt, <tag>_step_launches.txt
# Environment variables (PRD_*) are inherited by bench.py: A/B arms are separate invocations with different tags.
TAG=${1:-step}; shift
export TMPDIR=/tmp
rm -rf gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o t -- python3 bench.py --no-cpu-baseline --no-traffic --steps 20 --warmup 2 "$@" > gpurun_out/${TAG}_trace.log 2>&1
DB=$(find gpurun_out/${TAG}_trace -name "*.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/${TAG}_kernel_stats.txt > /dev/null
python tools/step_profile.py $DB > gpurun_out/${TAG}_step_breakdown.txt
python tools/step_launches.py $DB > gpurun_out/${TAG}_step_launches.txt
rm -rf gpurun_out/${TAG}_trace
cat gpurun_out/${TAG}_step_breakdown.txt
