#!/bin/bash
cd "$(dirname "$0")/.."
for ml in 7 6; do for m in 12 13; do for mg in 24 40; do
  echo "== min_low=$ml m=$m max_gates=$mg"
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --max-gates $mg --tile-bits $m --min-low $ml 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ms/step %.1f  passes %d  trips %d  avg_launch_ms %.2f  physical %.0f GB/s  value %.0f' % (d['ms_per_step'], d['config']['fused_passes_per_step'], d['config']['lds_round_trips_per_step'], r['avg_launch_ms'], r['physical_GBs'], d['value']))"
done; done; done
