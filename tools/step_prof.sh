export TMPDIR=/tmp
rm -rf gpurun_out/r03_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_trace -o t -- python3 bench.py --no-cpu-baseline --no-traffic --steps 20 --warmup 2 > gpurun_out/r03_trace.log 2>&1
DB=$(find gpurun_out/r03_trace -name "*.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/r03_kernel_stats.txt > /dev/null
python tools/step_profile.py $DB > gpurun_out/r03_step_breakdown.txt
rm -rf gpurun_out/r03_trace
cat gpurun_out/r03_step_breakdown.txt
