#!/bin/bash
# round 4, last session: reverse sweeps of states below a tile on the zero-padded (psi, lambda) pair -- tests, then the
# reference's gradient benchmark circuits and tools/bench_small.py with the switch on / off.  One gpurun call.
cd "$(dirname "$0")/../.."
out=gpurun_out/small_sweep
mkdir -p $out
timeout 600 python -m pytest tests/test_circuit_gpu.py -q -m gpu -x -k "smaller_than_a_tile or hip_graph_capture or fused_reverse_sweep" 2>&1 | tail -5 > $out/tests.txt
timeout 400 python tools/bench_gradient_reference.py --no-hessian > $out/gradient_reference_on.txt 2>&1
timeout 400 python tools/bench_gradient_reference.py --no-hessian --no-small-fused-sweep > $out/gradient_reference_off.txt 2>&1
for cfg in "--n 8 --depth 20 --batch 64" "--n 4 --depth 10 --batch 256" "--n 10 --depth 20 --batch 16"; do
  timeout 300 python tools/bench_small.py $cfg >> $out/bench_small.txt 2>&1
done
cat $out/tests.txt; cat $out/gradient_reference_on.txt; cat $out/gradient_reference_off.txt; cat $out/bench_small.txt
