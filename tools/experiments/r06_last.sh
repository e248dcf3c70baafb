#!/bin/bash
# Round 6, last check of the final code: the GPU suite as the driver runs it, smoke, the default bench line
cd "$(dirname "$0")/../.."
out=gpurun_out/r06z; mkdir -p $out
( time python -m pytest tests -m gpu -x -q --durations=12 ) > $out/gpu_suite.txt 2>&1
tail -4 $out/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python -c "
import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['parity_checked'], d['roofline']['frac'], d['roofline']['frac_of_measured_ceiling'])"
