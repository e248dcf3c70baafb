#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
out=gpurun_out/r05; mkdir -p $out
python tools/bench_vmap.py 20 16 > $out/bench_vmap.txt 2>&1; tail -8 $out/bench_vmap.txt
python tools/bench_vmap.py 24 8 >> $out/bench_vmap.txt 2>&1; tail -8 $out/bench_vmap.txt
timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_circuit_gpu.py -x -q -m gpu -k "fullsize or config4 or config5 or fused_node or torch_func or zero or pin" --durations=8 2>&1 | tail -25
for r in 0 1; do
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --virtual-bits 2 --steps 5 --warmup 1 > $out/rehearse_placed_r${r}_v2.json 2> $out/rehearse_placed_r${r}_v2.err
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --virtual-bits 0 --steps 5 --warmup 1 > $out/rehearse_placed_r${r}_v0.json 2> $out/rehearse_placed_r${r}_v0.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05/rehearse_placed*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['compute_ms_per_step'],1), d['fused_launches_per_step'], d['schedule'], round(d['modelled_step_ms'],1))
PY
