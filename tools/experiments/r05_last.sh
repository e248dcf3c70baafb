#!/bin/bash
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/last && out=gpurun_out/last
timeout 1500 python -m pytest tests -m gpu -x -q -k "wave or two_target or ansatz or golden or circuits_match or gate" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
python tools/experiments/r05_eager_ab.py 2>&1 | grep -v amdgpu | head -4
python tools/bench_two_qubit_rotations.py 2>&1 | grep -v amdgpu | tee $out/bench_two_qubit_rotations.txt
python tools/bench_two_qubit_rotations.py --n 22 --layers 8 2>&1 | grep -v amdgpu | tee -a $out/bench_two_qubit_rotations.txt
