#!/bin/bash
# rocprofv3 --kernel-trace --stats of the n = 28 training step (tools/bench_train.py, adjoint mode): the kernels of a step by total time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/train_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/train_trace/raw -o train -- python $R/tools/bench_train.py --n 28 --depth 40 --modes adjoint --reps 5 > $R/gpurun_out/train_trace/bench.txt 2>&1
f=$(find $R/gpurun_out/train_trace/raw -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/train_trace/train_step_kernel_stats.csv
head -12 $R/gpurun_out/train_trace/train_step_kernel_stats.csv | cut -c1-220; grep adjoint $R/gpurun_out/train_trace/bench.txt | cut -c1-250
rm -rf $R/gpurun_out/train_trace/raw
