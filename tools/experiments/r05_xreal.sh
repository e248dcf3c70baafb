#!/bin/bash
# DQ_MODE_XREAL: the X-shaped real bodies for channel superoperators.  Parity tests, then tools/bench_density.py.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/xreal && out=gpurun_out/xreal
timeout 1800 python -m pytest tests -m gpu -x -q -k "wave or density or den_mat or channel or noise" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
for rep in 1 2; do
  for flags in "--real-bodies" ""; do
    echo "# bench_density.py $flags" >> $out/density.txt
    timeout 600 python tools/bench_density.py --n 14 --depth 10 $flags 2>&1 | grep -v amdgpu.ids >> $out/density.txt
    timeout 600 python tools/bench_density.py --n 14 --depth 10 --no-zero-state $flags 2>&1 | grep -v amdgpu.ids >> $out/density.txt
    timeout 600 python tools/bench_density.py --n 15 --depth 6 --no-zero-state $flags 2>&1 | grep -v amdgpu.ids >> $out/density.txt
  done
done
cat $out/density.txt
