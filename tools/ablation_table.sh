#!/bin/bash
# Where a pass of the headline circuit spends its time: the same bench with parts of the fused kernel removed
# (tools/ablate.sh builds the variants; their RESULTS are wrong, only the time counts), with fewer workgroups per CU
# (LDS padding) and without the next-tile prefetch.  usage (GPU box, after tools/ablate.sh here): bash tools/ablation_table.sh
cd "$(dirname "$0")/.."
run() { tag=$1; shift; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-compare "$@" > /tmp/abl_$tag.json 2> /tmp/abl_$tag.err
  python -c "import json;d=json.load(open('/tmp/abl_$tag.json'));r=d['roofline'];print('%-34s %7.1f ms/step  %6.2f ms/pass  %5.0f GB/s  %.3f of peak' % ('$tag', d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac']))" || tail -2 /tmp/abl_$tag.err; }
echo "# headline workload (n=28, depth 40, c64, batch 16), 3 steps each, same box"
run full_kernel
run no_prefetch_one_tile_per_wg --tiles-per-wg 1
for v in nogates nolds nobar nogates_nolds; do
  [ -f deepquantum_amd/csrc/build/ablate/libdqhip_$v.so ] && DQHIP_LIBRARY=$PWD/deepquantum_amd/csrc/build/ablate/libdqhip_$v.so run ablated_$v
done
DQ_LDS_PAD_KB=24 run one_workgroup_per_cu
run tile12_four_workgroups_per_cu --tile-bits 12
DQ_LDS_PAD_KB=20 run tile12_three_workgroups_per_cu --tile-bits 12
DQ_LDS_PAD_KB=40 run tile12_two_workgroups_per_cu --tile-bits 12
run unmerged_gates --no-merge
run fixed_low_bits --no-free-low
run in_place_no_permuted_stores --no-permute-store
