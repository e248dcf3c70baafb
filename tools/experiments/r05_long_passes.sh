#!/bin/bash
# ABI 24: reverse-sweep passes of up to 104 gates + reductions, records in device memory.  Parity tests, then the training
# step with the old cap (72), with what the kernel-argument segment holds (80) and with the new default (104).
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/long_passes && out=gpurun_out/long_passes
timeout 1800 python -m pytest tests -m gpu -x -q -k "sweep or grad or train or hessian or adjoint or wave or graph" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
for rep in 1 2; do
  for mg in 72 80 0; do
    echo "DQ_MAX_GATES=$mg (0: the default, 104 for sweeps)" >> $out/train.txt
    e=""; [ $mg != 0 ] && e="DQ_MAX_GATES=$mg"
    env $e timeout 600 python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids >> $out/train.txt
    env $e timeout 600 python tools/bench_train.py --n 27 --depth 40 --modes adjoint --dtype c128 2>&1 | grep -v amdgpu.ids >> $out/train.txt
  done
done
cat $out/train.txt
