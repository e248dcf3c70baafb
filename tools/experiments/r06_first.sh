#!/bin/bash
# Round 6, first GPU call: the box, the last pass's knobs (tiles per wave of the <Z..Z>-reducing instantiation, XCD mapping
# with several tiles per wave), the bench line with the measured ceiling, the tightened pin test, the whole suite's durations.
cd "$(dirname "$0")/../.."
out=gpurun_out/r06; mkdir -p $out
{ rocm-smi --showcomputepartition 2>&1 | grep -i partition; python -c "import torch;print('devices',torch.cuda.device_count())"; } > $out/box.txt 2>&1
for v in "default:" "expz_tpw1:DQ_WAVE_EXPZ_TPW=1" "expz_tpw2_xcd:DQ_WAVE_EXPZ_TPW=2 DQ_WAVE_XCD_TPW=1" "expz_tpw4_xcd:DQ_WAVE_EXPZ_TPW=4 DQ_WAVE_XCD_TPW=1"; do
  tag=${v%%:*}; envs=${v#*:}
  echo "## $tag ($envs)" >> $out/last_pass_knobs.txt
  env $envs python tools/dump_passes.py 2>&1 | grep -E "pass 1[5-8]|total" >> $out/last_pass_knobs.txt
done
python bench.py --steps 10 --warmup 3 > $out/bench_first.json 2> $out/bench_first.err
( time python -m pytest tests/test_circuit_gpu.py -k "config3_pin" -x -q ) > $out/pin_test.txt 2>&1
( time python -m pytest tests -m gpu -x -q --durations=60 ) > $out/gpu_suite_first.txt 2>&1
tail -5 $out/gpu_suite_first.txt
