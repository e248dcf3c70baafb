#!/bin/bash
# Round 6: the whole GPU suite as the driver runs it (durations), the default bench line, the training step.
cd "$(dirname "$0")/../.."
out=gpurun_out/r06; mkdir -p $out
( time python -m pytest tests -m gpu -x -q --durations=25 ) > $out/gpu_suite.txt 2>&1
tail -4 $out/gpu_suite.txt
python bench.py > $out/bench_default_mid.json 2> $out/bench_default_mid.err
python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids > $out/train_n28.txt
DQ_WAVE_XCD_TPW=0 python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids >> $out/train_n28.txt
cat $out/train_n28.txt
