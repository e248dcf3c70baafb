#!/bin/bash
# Where a pass of the headline circuit spends its time: the same bench with one feature of the wave-tile kernel / the
# planner switched off at a time.  usage (GPU box): bash tools/ablation_table.sh
cd "$(dirname "$0")/.."
run() { tag=$1; shift; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare --no-sweep "$@" > /tmp/abl_$tag.json 2> /tmp/abl_$tag.err
  python -c "import json;d=json.load(open('/tmp/abl_$tag.json'));r=d['roofline'];print('%-44s %7.1f ms/step  %2d passes  %6.2f ms/pass  %5.0f GB/s  %.3f of peak' % ('$tag', d['ms_per_step'], d['config']['fused_passes_per_step'], r['avg_launch_ms'], r['achieved'], r['frac']))" || tail -2 /tmp/abl_$tag.err; }
echo "# headline workload (n=28, depth 40, c64, batch 16), 3 steps each, same box"
run full_kernel
run every_pass_moves_the_whole_state --no-zero-state
DQ_WAVE_NT=0 run plain_loads_and_stores
DQ_WAVE_NT=1 run streaming_loads_only
DQ_WAVE_NT=2 run streaming_stores_only
DQ_WAVE_TILE_ORDER=read run tile_numbers_in_read_order
DQ_WAVE_TILE_ORDER=read DQ_WAVE_NT=0 run read_order_and_plain_accesses
DQ_WAVE_XCD=0 run tile_groups_round_robin_over_the_xcds
DQ_WAVE_LDS_KB=54 run two_workgroups_per_cu_2_waves_per_simd
DQ_WAVE_LDS_KB=80 run one_workgroup_per_cu_1_wave_per_simd
run unmerged_gates --no-merge
run fixed_low_bits --no-free-low
run in_place_no_permuted_stores --no-permute-store
echo "# skeleton of the wave-tile kernel: synthetic passes with next to no gates (tools/experiments/nt_ab.py), ms per pass"
for nt in 0 3; do DQ_WAVE_NT=$nt python tools/experiments/nt_ab.py 2>&1 | grep "NT="; done
