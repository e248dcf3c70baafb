#!/bin/bash
# Round 6, final refresh: everything profiles/r06/ holds that comes from the GPU box, with the final code.
cd "$(dirname "$0")/../.."
out=gpurun_out/r06f; mkdir -p $out
bash tools/refresh_profiles.sh r06 > $out/refresh.log 2>&1
# all eight ranks of the strong-scaling job, sliced (default) and unsliced, + the anchor
timeout 600 python bench.py --strong --steps 5 --warmup 1 --no-cpu-baseline --no-compare --no-sweep > $out/anchor_n31.json 2> $out/anchor_n31.err
for r in 0 1 2 3 4 5 6 7; do
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 > $out/rehearse_strong_r${r}.json 2> $out/rehearse_strong_r${r}.err
  timeout 600 python bench.py --gpus 8 --strong --rehearse-rank $r --steps 5 --warmup 1 --slice-exchange 0 > $out/rehearse_strong_r${r}_unsliced.json 2> $out/rehearse_strong_r${r}_unsliced.err
done
# config 4 (n = 32 on four ranks) and the weak series' rank 0 (n = 31 on eight, batch 16), rehearsed
for r in 0 1; do timeout 600 python bench.py --gpus 4 --config 4 --rehearse-rank $r --steps 5 --warmup 1 > $out/rehearse_config4_r${r}.json 2> $out/rehearse_config4_r${r}.err; done
timeout 900 python bench.py --gpus 8 --rehearse-rank 0 --steps 3 --warmup 1 > $out/rehearse_weak_n31_r0.json 2> $out/rehearse_weak_n31_r0.err
# functional: two gloo ranks sharing the GPU, sliced exchange forced, strong-scaling circuit at n = 24
python bench.py --gpus 2 --backend gloo --steps 1 --warmup 1 --nqubit 23 --strong --slice-exchange 2 --no-cpu-baseline --no-sweep 2>&1 | grep '^{' | tail -1 > $out/two_ranks_sliced_functional.json
( time python -m pytest tests -m gpu -x -q --durations=15 ) > $out/gpu_suite.txt 2>&1
tail -3 $out/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
