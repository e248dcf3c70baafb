#!/bin/bash
# Functional check of bench.py's N > 1 paths on a box with ONE GPU: two ranks share it, gloo carries the
# collectives (host staged).  Not a performance number.  usage: tools/bench_two_ranks_one_gpu.sh [bench args]
cd "$(dirname "$0")/.."
port=$((20000 + RANDOM % 20000))
for r in 0 1; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=$port WORLD_SIZE=2 RANK=$r LOCAL_RANK=0 \
    python bench.py --gpus 2 --backend gloo --steps 1 --warmup 1 "$@" > /tmp/bench_rank$r.log 2>&1 &
  pids[$r]=$!
done
rc=0
for r in 0 1; do wait ${pids[$r]} || rc=$?; done
grep -h '^{' /tmp/bench_rank0.log || tail -20 /tmp/bench_rank0.log /tmp/bench_rank1.log
exit $rc
