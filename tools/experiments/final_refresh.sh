mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/gpu_suite_r03.txt
timeout 2400 bash tools/refresh_profiles.sh r03 > gpurun_out/refresh.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.txt 2>&1
tail -3 gpurun_out/gpu_suite_r03.txt; cat gpurun_out/prof/r03/bench_default.json
