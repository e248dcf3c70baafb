#!/bin/bash
# Experiment: is instruction fetch part of what a heavy pass costs?  (the kernel's code is ~250 KB, a pass walks 25-35
# handler bodies of 0.4-2 KB each; the instruction cache is shared by two CUs.)  usage: gpurun -- bash tools/experiments/r04_icache.sh
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
root=$PWD/gpurun_out/prof/icache
rm -rf "$root"; mkdir -p "$root"
BENCH="python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-compare --no-sweep"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o -E "\b(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_LEVEL|INSTS_BRANCH|INSTS_CBRANCH|CBRANCH)[A-Z_]*)\b" | sort -u > "$root/avail.txt"
cat "$root/avail.txt"
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "wave_pass" --pmc $set -d "$root/$tag" -o pmc -- $BENCH > /dev/null 2> "$root/$tag.err"
  tail -2 "$root/$tag.err"
done
cd - > /dev/null
python - "$root" <<'PY' | tee "$root/summary.txt"
import csv, glob, os, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'wave_pass' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in acc.items():
        v.sort()
        full = v[len(v) // 2:]          # (the upper half of the launches: full passes)
        print('%-30s launches %3d  mean %.4g  mean of the upper half %.4g  max %.4g' % (c, len(v), sum(v) / len(v), sum(full) / len(full), v[-1]))
PY
find "$root" -name '*.db' -delete
find "$root" -name '*kernel_trace.csv' -size +2M -delete
find "$root" -name '*counter_collection.csv' -size +4M -delete
