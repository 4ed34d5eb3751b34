# Kernel stats and step breakdown of the long-row configurations (GPU box, through gpurun):  bash tools/collect_long_rows.sh
export TMPDIR=/tmp
for cfg in "768 1 n769" "1000 24 n1024"; do
  set -- $cfg
  rm -rf gpurun_out/r03_$3
  rocprofv3 --kernel-trace --stats -d gpurun_out/r03_$3 -o t -- python3 bench.py --no-cpu-baseline --no-traffic --residues $1 --atoms $2 --steps 10 --warmup 2 > gpurun_out/r03_$3.log 2>&1
  DB=$(find gpurun_out/r03_$3 -name "*.db" | head -1)
  python tools/rocprof_summary.py $DB gpurun_out/r03_$3_kernel_stats.txt > /dev/null
  python tools/step_profile.py $DB > gpurun_out/r03_$3_step_breakdown.txt
  rm -rf gpurun_out/r03_$3
  head -8 gpurun_out/r03_$3_step_breakdown.txt
done
