#!/bin/bash
# the round's evidence in one gpurun call: the GPU suite, refresh_profiles.sh <tag>, smoke.  usage: bash tools/experiments/final_refresh.sh r04
cd "$(dirname "$0")/../.."
tag=${1:-r05}
mkdir -p gpurun_out
timeout 3000 bash tools/refresh_profiles.sh $tag > gpurun_out/refresh.log 2>&1
timeout 3000 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -22 > gpurun_out/prof/$tag/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/prof/$tag/smoke.txt 2>&1
tail -3 gpurun_out/prof/$tag/gpu_suite.txt; tail -2 gpurun_out/prof/$tag/smoke.txt; cat gpurun_out/prof/$tag/bench_default.json | cut -c1-600
