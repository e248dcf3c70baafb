#!/bin/bash
# round 4, experiment 3 (one box): tiles per wave with the XCD mapping kept; state size / spread of the writes
cd "$(dirname "$0")/../.."
out=gpurun_out/r04d; mkdir -p $out
run() { tag=$1; shift; python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-compare --no-sweep "$@" > /tmp/abl_$tag.json 2> /tmp/abl_$tag.err
  python -c "import json;d=json.load(open('/tmp/abl_$tag.json'));r=d['roofline'];print('%-44s %7.1f ms/step  %2d passes  %6.2f ms/pass  %5.0f GB/s  %.3f of peak parity %s' % ('$tag', d['ms_per_step'], d['config']['fused_passes_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'], d.get('parity_checked')))" || tail -2 /tmp/abl_$tag.err; }
{
run base
for t in 2 4 8 16 64; do DQ_WAVE_TPW=$t run tpw_$t; done
run base_again
run n26_b64 --nqubit 26 --batch 64
run n24_b256 --nqubit 24 --batch 256
run n30_b4 --nqubit 30 --batch 4
} > $out/exp_tpw.txt 2>&1
cat $out/exp_tpw.txt
python -m pytest tests/test_distributed_gpu.py -q -x > $out/dist_gpu.txt 2>&1; tail -3 $out/dist_gpu.txt
for a in "" "--strong"; do bash tools/bench_two_ranks_one_gpu.sh --nqubit 24 --depth 10 --batch 4 --no-cpu-baseline --no-sweep $a 2>&1 | tail -1; done > $out/two_ranks.txt 2>&1
python - <<'PY'
import json
for line in open('gpurun_out/r04d/two_ranks.txt'):
    try: d = json.loads(line)
    except Exception: print(line[:300]); continue
    print({k: d.get(k) for k in ('ms_per_step', 'ms_per_step_with_restore')}, d['config'].get('collectives_per_step'), d['config'].get('exchange_per_step'))
    for r in d['config']['remap_timings']['remaps']: print('   ', r)
PY
