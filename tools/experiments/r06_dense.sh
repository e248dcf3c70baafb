#!/bin/bash
# Round 6: dense gates k = 6, 7 with U stationary in LDS (DQ_DENSE67 = 0 old / 1 all rows per wave / 2 half), correctness first
cd "$(dirname "$0")/../.."
out=gpurun_out/r06; mkdir -p $out
for v in 1 2 0; do
  echo "## DQ_DENSE67=$v" >> $out/dense67.txt
  DQ_DENSE67=$v timeout 600 python -m pytest tests/test_kernels_gpu.py -k "dense" -x -q 2>&1 | tail -2 >> $out/dense67.txt
  DQ_DENSE67=$v timeout 600 python tools/bench_dense.py 2>&1 | grep -E "^ *(x64|128) +[0-9]+ +[567] " >> $out/dense67.txt
done
for wg in 1 3 4; do
  echo "## DQ_DENSE67=1 DQ_DENSE67_WG=$wg" >> $out/dense67.txt
  DQ_DENSE67=1 DQ_DENSE67_WG=$wg timeout 600 python tools/bench_dense.py 2>&1 | grep -E "^ *(x64|128) +[0-9]+ +[67] " >> $out/dense67.txt
  echo "## DQ_DENSE67=2 DQ_DENSE67_WG=$wg" >> $out/dense67.txt
  DQ_DENSE67=2 DQ_DENSE67_WG=$wg timeout 600 python tools/bench_dense.py 2>&1 | grep -E "^ *(x64|128) +[0-9]+ +[67] " >> $out/dense67.txt
done
cat $out/dense67.txt
