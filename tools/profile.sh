#!/bin/bash
# rocprofv3 recipe for the headline bench (run on the GPU box via gpurun).  Writes under gpurun_out/prof/.
# usage: tools/profile.sh <tag> [bench args...]
cd "$(dirname "$0")/.."
tag=${1:-r01}; shift
export TMPDIR=/tmp
root=$PWD/gpurun_out/prof/$tag
rm -rf "$root"; mkdir -p "$root"
BENCH="python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-compare $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/trace" -o trace -- $BENCH > "$root/bench_trace.json" 2> "$root/trace.err"
rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "fused_pass|wave_pass" --pmc FETCH_SIZE -d "$root/pmc_fetch" -o pmc -- $BENCH > /dev/null 2> "$root/pmc_fetch.err"
rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "fused_pass|wave_pass" --pmc WRITE_SIZE -d "$root/pmc_write" -o pmc -- $BENCH > /dev/null 2> "$root/pmc_write.err"
rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "fused_pass|wave_pass" --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d "$root/pmc_sq1" -o pmc -- $BENCH > /dev/null 2> "$root/pmc_sq1.err"
rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "fused_pass|wave_pass" --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$root/pmc_sq2" -o pmc -- $BENCH > /dev/null 2> "$root/pmc_sq2.err"
cd - > /dev/null
python tools/summarize_prof.py "$root" | tee "$root/summary.txt"
# keep the bulky raw traces out of the merged output (64 MiB cap): only stats + summaries
find "$root" -name '*.db' -delete
find "$root" -name '*kernel_trace.csv' -size +2M -delete
find "$root" -name '*counter_collection.csv' -size +4M -delete
