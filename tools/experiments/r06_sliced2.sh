#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r06; mkdir -p $out
( time python -m pytest tests/test_distributed_gpu.py -k "sliced" -x -q ) > $out/sliced_dist_tests.txt 2>&1; grep -E "passed|failed" $out/sliced_dist_tests.txt
for r in 0 1 2 3 4 5 6 7; do
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 > $out/rehearse4_r${r}.json 2> $out/rehearse4_r${r}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06/rehearse4_r*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'failed',e, open(f.replace('.json','.err')).read()[-600:]); continue
    print(f.split('/')[-1], 'compute %.1f launches %.0f sum %.1f exposed %.1f modelled %.1f (+contention %.1f)'%(d['compute_ms_per_step'],d['fused_launches_per_step'],d['fused_launch_ms_sum_per_step'],d['wire_model']['wire_ms_per_step_exposed_model'],d['modelled_step_ms'],d['modelled_step_ms_with_hbm_contention']), [(w['launches_of_the_last_pass'],w['launches_of_the_first_pass_behind']) for w in d['wire_model']['remaps']])
PY
