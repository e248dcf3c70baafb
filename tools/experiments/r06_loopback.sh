#!/bin/bash
# Round 6: what the hidden wire costs the passes beside it, MEASURED (loopback copies of the slices on the exchange stream), and a
# third slice bit
cd "$(dirname "$0")/../.."
out=gpurun_out/r06l; mkdir -p $out
for r in 0 1 6; do
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 > $out/r${r}_sliced.json 2> $out/r${r}_sliced.err
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 --rehearse-loopback > $out/r${r}_sliced_loopback.json 2> $out/r${r}_sliced_loopback.err
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 --slice-exchange 3 > $out/r${r}_three_bits.json 2> $out/r${r}_three_bits.err
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 --slice-exchange 3 --rehearse-loopback > $out/r${r}_three_bits_loopback.json 2> $out/r${r}_three_bits_loopback.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06l/r*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'failed',e, open(f.replace('.json','.err')).read()[-800:]); continue
    print('%-28s compute %.1f launches %.0f exposed %.1f modelled %.1f (+model contention %.1f)'%(f.split('/')[-1],d['compute_ms_per_step'],d['fused_launches_per_step'],d['wire_model']['wire_ms_per_step_exposed_model'],d['modelled_step_ms'],d['modelled_step_ms_with_hbm_contention']), [(w['launches_of_the_last_pass'],w['launches_of_the_first_pass_behind']) for w in d['wire_model']['remaps']])
PY
