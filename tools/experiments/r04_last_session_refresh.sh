#!/bin/bash
# round 4, last session: the evidence that changed, in one gpurun call -- rocprofv3 passes of the headline step (tools/profile.sh),
# the default bench line, launch-bound benchmarks, the reference's gradient / Hessian chart, training steps (regression check
# of the large sizes), smoke, the GPU suite.  usage: bash tools/experiments/r04_last_session_refresh.sh
cd "$(dirname "$0")/../.."
tag=r04b
root=$PWD/gpurun_out/prof/$tag
timeout 900 bash tools/profile.sh "$tag" > /dev/null 2>&1
cp "$root/traffic.json" "$root/traffic_n28_b16_c64.json" 2>/dev/null
find "$root/trace" -name '*kernel_stats.csv' -exec cp {} "$root/kernel_stats.csv" \;
cp "$root/bench_trace.json" "$root/bench_under_rocprof.json"
timeout 600 python bench.py --traffic-json "$root/traffic_n28_b16_c64.json" > "$root/bench_default_last_commit.json" 2> "$root/bench_default.err"
{
  for cfg in "--n 8 --depth 20 --batch 64" "--n 4 --depth 10 --batch 256" "--n 10 --depth 20 --batch 16"; do
    timeout 300 python tools/bench_small.py $cfg 2>&1 | grep -v amdgpu.ids
  done
} > "$root/bench_small.txt" 2>&1
timeout 900 python tools/bench_gradient_reference.py --trials 3 2>&1 | grep -v "amdgpu.ids\|UserWarning\|run_backward" > "$root/bench_gradient_reference.txt"
{
  timeout 300 python tools/bench_train.py --n 20 --depth 20 --modes adjoint 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/bench_train.py --n 24 --depth 20 --modes adjoint 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids
  timeout 300 python tools/bench_train.py --n 27 --depth 40 --modes adjoint --dtype c128 2>&1 | grep -v amdgpu.ids
} > "$root/training_steps.txt" 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$root/smoke.txt" 2>&1
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > "$root/gpu_suite.txt"
tail -3 "$root/gpu_suite.txt"; tail -2 "$root/smoke.txt"; cut -c1-400 "$root/bench_default_last_commit.json"; cat "$root/training_steps.txt"; tail -20 "$root/summary.txt"
