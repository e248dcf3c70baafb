#!/bin/bash
# round 4, experiment 2 (one box): gate cap per pass (VALU balance between passes); the reduced GRAD sums
cd "$(dirname "$0")/../.."
out=gpurun_out/r04c; mkdir -p $out
run() { tag=$1; shift; python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-compare --no-sweep "$@" > /tmp/abl_$tag.json 2> /tmp/abl_$tag.err
  python -c "import json;d=json.load(open('/tmp/abl_$tag.json'));r=d['roofline'];print('%-44s %7.1f ms/step  %2d passes  %6.2f ms/pass  %5.0f GB/s  %.3f of peak parity %s' % ('$tag', d['ms_per_step'], d['config']['fused_passes_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'], d.get('parity_checked')))" || tail -2 /tmp/abl_$tag.err; }
{
run base
for g in 44 48 52 56 60 64; do run max_gates_$g --max-gates $g; done
run base_again
} > $out/exp_gatecap.txt 2>&1
cat $out/exp_gatecap.txt
python -m pytest tests -m gpu -q -x -k "sweep or grad or train or adjoint or reductions" > $out/grad_tests.txt 2>&1; tail -3 $out/grad_tests.txt
{
python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids
DQ_REDUCED_GRAD=0 python tools/bench_train.py --n 28 --depth 40 --modes adjoint 2>&1 | grep -v amdgpu.ids
python tools/bench_train.py --n 24 --depth 20 --modes adjoint 2>&1 | grep -v amdgpu.ids
python tools/dump_sweep_passes.py 2>&1 | grep -v amdgpu.ids
} > $out/train.txt 2>&1
cat $out/train.txt
