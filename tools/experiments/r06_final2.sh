#!/bin/bash
# Round 6, last rehearsal: three slice bits (the default now), all eight ranks, exchanges elided and as loopback copies; + anchor
cd "$(dirname "$0")/../.."
out=gpurun_out/r06g; mkdir -p $out
timeout 600 python bench.py --strong --steps 5 --warmup 1 --no-cpu-baseline --no-compare --no-sweep > $out/anchor_n31.json 2> $out/anchor_n31.err
for r in 0 1 2 3 4 5 6 7; do
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 > $out/rehearse_strong_r${r}.json 2> $out/rehearse_strong_r${r}.err
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 --rehearse-loopback > $out/rehearse_strong_r${r}_loopback.json 2> $out/rehearse_strong_r${r}_loopback.err
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 --slice-exchange 0 > $out/rehearse_strong_r${r}_unsliced.json 2> $out/rehearse_strong_r${r}_unsliced.err
done
for r in 0 1; do timeout 600 python bench.py --gpus 4 --config 4 --rehearse-rank $r --steps 5 --warmup 1 > $out/rehearse_config4_r${r}.json 2> $out/rehearse_config4_r${r}.err; done
( time python -m pytest tests/test_distributed_gpu.py tests/test_fullsize_gpu.py -x -q ) > $out/dist_gpu_tests.txt 2>&1; grep -E "passed|failed" $out/dist_gpu_tests.txt
python - <<'PY'
import json,glob
a=json.load(open('gpurun_out/r06g/anchor_n31.json')); print('anchor', a['ms_per_step'])
for f in sorted(glob.glob('gpurun_out/r06g/rehearse_strong_r*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'failed',e, open(f.replace('.json','.err')).read()[-800:]); continue
    print('%-36s compute %.1f launches %.0f exposed %.1f modelled %.1f (+model contention %.1f)'%(f.split('/')[-1],d['compute_ms_per_step'],d['fused_launches_per_step'],d['wire_model']['wire_ms_per_step_exposed_model'],d['modelled_step_ms'],d['modelled_step_ms_with_hbm_contention']), [(w['launches_of_the_last_pass'],w['launches_of_the_first_pass_behind']) for w in d['wire_model']['remaps']])
PY
