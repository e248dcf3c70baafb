#!/bin/bash
# DQ_FG_GRAD variant 4 (one sum per X rotation in a backward that records no graph): parity tests, then the training step
# of tools/bench_train.py with and without it (same box, same process order twice).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/one_sum && out=gpurun_out/one_sum
timeout 1500 python -m pytest tests -m gpu -x -q -k "sweep or grad or train or hessian or adjoint or wave" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
for rep in 1 2; do
  for t in 0 1; do
    echo "DQ_TERMINAL_GRAD=$t" >> $out/train.txt
    DQ_TERMINAL_GRAD=$t timeout 600 python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids >> $out/train.txt
    DQ_TERMINAL_GRAD=$t timeout 600 python tools/bench_train.py --n 27 --depth 40 --modes adjoint --dtype c128 2>&1 | grep -v amdgpu.ids >> $out/train.txt
  done
done
cat $out/train.txt
