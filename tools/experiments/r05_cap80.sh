#!/bin/bash
# more records per pass in the reverse sweep (the planner's cap of 72 gates + reductions -> 80, what the ABI holds today)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/cap80 && out=gpurun_out/cap80
for rep in 1 2; do
  for mg in 72 80; do
    echo "DQ_MAX_GATES=$mg" >> $out/train.txt
    DQ_MAX_GATES=$mg timeout 600 python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids >> $out/train.txt
    DQ_MAX_GATES=$mg timeout 600 python tools/bench_train.py --n 27 --depth 40 --modes adjoint --dtype c128 2>&1 | grep -v amdgpu.ids >> $out/train.txt
  done
done
cat $out/train.txt
