mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/gpu_suite_r03.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.txt 2>&1
python tools/experiments/bench_shard_helpers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/shard_helpers_and_reductions.txt
python bench.py --traffic-json profiles/r03/traffic_n28_b16_c64.json > gpurun_out/bench_last.json 2> /dev/null
tail -2 gpurun_out/gpu_suite_r03.txt; tail -1 gpurun_out/smoke.txt; python -c "
import json; d=json.load(open('gpurun_out/bench_last.json')); print(d['ms_per_step'], d['roofline']['frac'], d['parity_checked'])"
