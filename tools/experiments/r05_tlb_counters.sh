#!/bin/bash
# (a counter set the hardware cannot collect in one pass makes rocprofv3 abort and then hang in its finaliser: small sets, and a
# hard timeout around every run)
# Address translation and L2 / fabric counters of the pass kernel, per launch (= per pass of the headline step), for the
# default schedule (permuted stores: a tile writes 256 runs of 128 bytes all over the state) and for in-place passes
# (--no-permute-store).  VERDICT r4 item 4(b): attribute the light pass's 0.70 vs the skeleton's 0.79 with counters.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
root=$PWD/gpurun_out/r05/tlb; rm -rf $root; mkdir -p $root
for variant in default inplace; do
  extra=""; [ $variant = inplace ] && extra="--no-permute-store"
  BENCH="python $PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-compare --no-parity $extra"
  i=0
  for set in "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS" "TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_MULTI_MISS" \
             "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY" "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" \
             "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_WRREQ_STALL TCC_TAG_STALL" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_TOO_MANY_EA_WRREQS_STALL"; do
    i=$((i+1))
    ( cd /tmp && timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv --kernel-include-regex "wave_pass" --pmc $set -d $root/${variant}_$i -o pmc -- $BENCH > /dev/null 2> $root/${variant}_$i.err )
  done
done
python - <<'PY'
import csv, glob, os, collections
root = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out/r05/tlb')
for variant in ('default', 'inplace'):
    per = collections.defaultdict(dict)       # dispatch id -> counter -> value
    dur = {}
    for f in glob.glob(f'{root}/{variant}_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            per[int(r['Dispatch_Id'])][r['Counter_Name']] = per[int(r['Dispatch_Id'])].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    ids = sorted(per)
    names = sorted({c for d in per.values() for c in d})
    print(f'## {variant}: {len(ids)} launches (warm-up step + timed step); per launch')
    print('launch ' + ' '.join(f'{n[-28:]:>28s}' for n in names))
    for k, i in enumerate(ids):
        print(f'{k:6d} ' + ' '.join(f'{per[i].get(n, float("nan")):28.4g}' for n in names))
PY
