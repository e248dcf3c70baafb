#!/bin/bash
cd "$(dirname "$0")/../.."
out=gpurun_out/r06h; mkdir -p $out
timeout 600 python bench.py --strong --steps 5 --warmup 1 --no-cpu-baseline --no-compare --no-sweep > $out/anchor_n31.json 2> $out/anchor_n31.err
for r in 0 1 2 3 4 5 6 7; do
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 > $out/rehearse_strong_r${r}.json 2> $out/rehearse_strong_r${r}.err
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 --rehearse-loopback > $out/rehearse_strong_r${r}_loopback.json 2> $out/rehearse_strong_r${r}_loopback.err
done
python - <<'PY'
import json,glob
a=json.load(open('gpurun_out/r06h/anchor_n31.json')); print('anchor', a['ms_per_step'])
for f in sorted(glob.glob('gpurun_out/r06h/rehearse_strong_r*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'failed',e, open(f.replace('.json','.err')).read()[-800:]); continue
    print('%-36s compute %.1f launches %.0f exposed %.1f modelled %.1f'%(f.split('/')[-1],d['compute_ms_per_step'],d['fused_launches_per_step'],d['wire_model']['wire_ms_per_step_exposed_model'],d['modelled_step_ms']), [(w['launches_of_the_last_pass'],w['launches_of_the_first_pass_behind']) for w in d['wire_model']['remaps']])
PY
