#!/bin/bash
# round 4, experiment 1 (one box): priority / store-interleave switches of the wave kernel, 256-byte runs everywhere,
# per-pass dump; the new tools
cd "$(dirname "$0")/../.."
out=gpurun_out/r04b; mkdir -p $out
run() { tag=$1; shift; python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-compare --no-sweep "$@" > /tmp/abl_$tag.json 2> /tmp/abl_$tag.err
  python -c "import json;d=json.load(open('/tmp/abl_$tag.json'));r=d['roofline'];print('%-44s %7.1f ms/step  %2d passes  %6.2f ms/pass  %5.0f GB/s  %.3f of peak parity %s' % ('$tag', d['ms_per_step'], d['config']['fused_passes_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'], d.get('parity_checked')))" || tail -2 /tmp/abl_$tag.err; }
{
run base
DQ_WAVE_EXP=1 run prio
DQ_WAVE_EXP=2 run interleave
DQ_WAVE_EXP=3 run prio_interleave
run base_again
DQ_WAVE_EXP=1 run prio_again
run min_low_5 --min-low 5
} > $out/exp_ab.txt 2>&1
cat $out/exp_ab.txt
python tools/dump_passes.py > $out/dump_passes.txt 2>&1
python -m pytest tests/test_integration_stub_gpu.py -q > $out/integration_stub.txt 2>&1; tail -3 $out/integration_stub.txt
python tools/bench_gradient_reference.py --trials 3 > $out/bench_gradient_reference.txt 2>&1; cat $out/bench_gradient_reference.txt
