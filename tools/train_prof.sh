export TMPDIR=/tmp
rm -rf gpurun_out/tp_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/tp_trace -o t -- python3 tools/train_bench.py --steps 3 --warmup 2 > gpurun_out/tp_trace.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/tp_trace -name "*.db" | head -1) gpurun_out/r04b_train_kernel_stats.txt > /dev/null
rm -rf gpurun_out/tp_trace
head -50 gpurun_out/r04b_train_kernel_stats.txt
