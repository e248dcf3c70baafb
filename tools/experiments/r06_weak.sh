#!/bin/bash
# Round 6: the weak series' top line (n = 31 on 8 ranks, batch 16) and config 4 (n = 32 on 4 ranks), rehearsed with the final code
cd "$(dirname "$0")/../.."
out=gpurun_out/r06w; mkdir -p $out
for r in 0 5; do
  timeout 900 python bench.py --gpus 8 --rehearse-rank $r --steps 3 --warmup 1 > $out/rehearse_weak_n31_r${r}.json 2> $out/rehearse_weak_n31_r${r}.err
  timeout 900 python bench.py --gpus 8 --rehearse-rank $r --steps 3 --warmup 1 --no-defer-tail > $out/rehearse_weak_n31_r${r}_no_defer.json 2> $out/rehearse_weak_n31_r${r}_no_defer.err
done
for r in 0 1 2 3; do timeout 600 python bench.py --gpus 4 --config 4 --rehearse-rank $r --steps 5 --warmup 1 > $out/rehearse_config4_r${r}.json 2> $out/rehearse_config4_r${r}.err; done
timeout 600 python bench.py --config 4 --nqubit 32 --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-sweep > $out/one_gpu_n32.json 2> $out/one_gpu_n32.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06w/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'failed',e, open(f.replace('.json','.err')).read()[-500:]); continue
    if 'rehearsal' in d: print('%-40s compute %.1f launches %.0f exposed %.1f modelled %.1f'%(f.split('/')[-1],d['compute_ms_per_step'],d['fused_launches_per_step'],d['wire_model']['wire_ms_per_step_exposed_model'],d['modelled_step_ms']), d['schedule'].get('deferred_tails'))
    else: print(f.split('/')[-1], d['ms_per_step'], d.get('parity_checked'))
PY
