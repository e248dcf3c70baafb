#!/bin/bash
# the last GPU call of a round: the whole GPU suite, smoke, one default bench line and the training step, on the final code
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out && out=gpurun_out/final_suite.txt
timeout 2700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -14 > $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -3 >> $out
python bench.py --steps 5 --warmup 2 2>/dev/null | cut -c1-420 >> $out
for i in 1 2; do python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids >> $out; done
python tools/bench_train.py --n 27 --depth 40 --modes adjoint --dtype c128 2>&1 | grep -v amdgpu.ids >> $out
cat $out
