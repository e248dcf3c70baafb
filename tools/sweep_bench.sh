#!/bin/bash
# GPU-side parameter sweep of the fused kernel on the headline workload (scratch tool).
# usage: tools/sweep_bench.sh "<max_gates list>" "<m list>" [extra bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/sweep.log
: > $out
MG=${1:-"10 20 40"}; MS=${2:-"12"}; shift 2
for mg in $MG; do
  for m in $MS; do
    echo "== max_gates=$mg m=$m $*" >> $out
    timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --max-gates $mg --tile-bits $m "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ms/step %.1f  passes %d  trips %d  avg_launch_ms %.2f  physical %.0f GB/s  value %.0f' % (d['ms_per_step'], d['config']['fused_passes_per_step'], d['config']['lds_round_trips_per_step'], r['avg_launch_ms'], r['physical_GBs'], d['value']))" >> $out
  done
done
cat $out
