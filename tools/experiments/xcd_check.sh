timeout 1500 python -m pytest tests/test_wave_gpu.py tests/test_kernels_gpu.py tests/test_circuit_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in 0 1; do
  echo "DQ_WAVE_XCD=$v"
  DQ_WAVE_XCD=$v python bench.py --dtype c128 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' c128 n=28 b=8', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
  DQ_WAVE_XCD=$v python bench.py --strong --steps 3 --warmup 1 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' n=31 b=1', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
  DQ_WAVE_XCD=$v python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu | cut -c1-150
  DQ_WAVE_XCD=$v python tools/bench_config2.py 2>&1 | grep -v amdgpu
done
