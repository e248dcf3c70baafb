set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "marginal or z_strings or reductions" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_circuit_gpu.py -x -q -m gpu -k "marginal or measure or hamiltonian or expectation" 2>&1 | tail -3
timeout 300 python tools/experiments/bench_shard_helpers.py 2>&1 | grep -v permute | tee gpurun_out/shard_helpers_new.txt

