#!/bin/bash
# Round 6, second rehearsal: lazy reset + first exchange without wire or memset; the sharded GPU tests with the real kernels
cd "$(dirname "$0")/../.."
out=gpurun_out/r06; mkdir -p $out
( time python -m pytest tests/test_distributed_gpu.py tests/test_fullsize_gpu.py -x -q ) > $out/dist_gpu_tests.txt 2>&1
tail -4 $out/dist_gpu_tests.txt
for r in 0 1 2 3 4 5 6 7; do
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 > $out/rehearse2_r${r}.json 2> $out/rehearse2_r${r}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/rehearse2_r*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'failed',e); continue
    print(f.split('/')[-1], 'v',d['virtual_rank_bits'],'compute %.1f (events %.1f) launches %.0f sum %.1f wire_exposed %.1f modelled %.1f'%(d['compute_ms_per_step'],d['compute_ms_per_step_hip_events_median'],d['fused_launches_per_step'],d['fused_launch_ms_sum_per_step'],d['wire_model']['wire_ms_per_step_exposed_model'],d['modelled_step_ms']), d['schedule'])
PY
