#!/bin/bash
# the software-pipelined pass kernel (one wave per SIMD, tile images in the accumulation registers): parity where it is forced
# everywhere it can run (DQ_WAVE_PF=2), then the headline with and without it
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/pf && out=gpurun_out/pf
DQ_WAVE_PF=2 timeout 900 python -m pytest tests/test_wave_gpu.py -x -q -k "match_oracle or permuted or batched" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
for pf in 0 1 0 1; do
  echo "DQ_WAVE_PF=$pf" | tee -a $out/bench.txt
  DQ_WAVE_PF=$pf timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(d['ms_per_step'], 'ms/step; full launch', r.get('full_launch_avg_ms'), 'ms; frac', r['frac'], 'parity', d.get('parity_checked'))
" | tee -a $out/bench.txt
done
