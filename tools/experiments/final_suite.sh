mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/gpu_suite_r03.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.txt 2>&1
timeout 600 python tools/soak.py 9000 40 2>&1 | tail -2 > gpurun_out/soak_9000.txt
{ python tools/bench_config2.py; python tools/bench_config2.py --n 20; python tools/bench_small.py; python tools/bench_density.py; python tools/bench_expect.py; } 2>&1 | grep -v amdgpu.ids > gpurun_out/secondary_after_steady.txt
python bench.py --traffic-json profiles/r03/traffic_n28_b16_c64.json > gpurun_out/bench_last.json 2> /dev/null
tail -2 gpurun_out/gpu_suite_r03.txt; tail -1 gpurun_out/smoke.txt; cat gpurun_out/soak_9000.txt; cat gpurun_out/secondary_after_steady.txt; python -c "
import json; d=json.load(open('gpurun_out/bench_last.json')); print(d['ms_per_step'], d['roofline']['frac'], d['parity_checked'])"
