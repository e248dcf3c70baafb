#!/bin/bash
# Round 6: the compute half of the strong-scaling job (n = 34 on 8 ranks) for ALL EIGHT ranks on one GPU (every exchange
# left out), v chosen by the library's dry-run model, with and without the first exchange over the wire; the one-GPU anchor.
cd "$(dirname "$0")/../.."
out=gpurun_out/r06; mkdir -p $out
timeout 600 python bench.py --strong --steps 5 --warmup 1 --no-cpu-baseline --no-compare --no-sweep > $out/anchor_n31.json 2> $out/anchor_n31.err
for r in 0 1 2 3 4 5 6 7; do
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 > $out/rehearse_r${r}.json 2> $out/rehearse_r${r}.err
done
for r in 0 3; do
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 --no-local-first-exchange > $out/rehearse_r${r}_wire_first.json 2> $out/rehearse_r${r}_wire_first.err
done
python - <<'PY'
import json,glob
a=json.load(open('gpurun_out/r06/anchor_n31.json'))
print('anchor n=31 one GPU: %.1f ms/step'%a['ms_per_step'])
for f in sorted(glob.glob('gpurun_out/r06/rehearse_r*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'failed',e); continue
    print(f.split('/')[-1], 'v',d['virtual_rank_bits'],'compute %.1f (events %.1f) launches %.0f sum %.1f wire_exposed %.1f modelled %.1f'%(d['compute_ms_per_step'],d['compute_ms_per_step_hip_events_median'],d['fused_launches_per_step'],d['fused_launch_ms_sum_per_step'],d['wire_model']['wire_ms_per_step_exposed_model'],d['modelled_step_ms']), d['schedule'])
PY
