#!/bin/bash
# Round 5, first look: the rehearsal of the strong-scaling pair's ranks, the one-GPU anchor, and the sharded configs at
# full size as gloo ranks sharing the one GPU (functional + parity; not a performance figure).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
out=gpurun_out/r05; mkdir -p $out
t() { local name=$1; shift; local t0=$(date +%s.%N); timeout 1200 "$@" > $out/$name.json 2> $out/$name.err; local rc=$?
      echo "$name rc=$rc $(echo "$(date +%s.%N) - $t0" | bc) s"; }
t anchor_n31 python bench.py --strong --steps 5 --warmup 1 --no-cpu-baseline --no-compare
for r in 0 1; do for v in 2 0; do
  t rehearse_strong_r${r}_v$v python bench.py --gpus 8 --strong --rehearse-rank $r --virtual-bits $v --steps 5 --warmup 1
done; done
t config4_n32_w4_v0 python bench.py --gpus 4 --backend gloo --config 4 --steps 1 --warmup 0 --virtual-bits 0 --no-cpu-baseline
t config4_n32_w4_v2 python bench.py --gpus 4 --backend gloo --config 4 --steps 1 --warmup 0 --virtual-bits 2 --no-cpu-baseline
t weak_n29_w2 python bench.py --gpus 2 --backend gloo --steps 1 --warmup 0 --no-cpu-baseline
t config5_n33_w8 python bench.py --gpus 8 --backend gloo --config 5 --nqubit 30 --steps 1 --warmup 0 --virtual-bits 2 --no-cpu-baseline --qaoa-nqubit 31
grep -h -o '"parity_checked": [a-z]*' $out/*.json
tail -3 $out/*.err | tail -40
