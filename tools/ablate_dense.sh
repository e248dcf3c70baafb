#!/bin/bash
# Timing experiments on the staged dense-gate kernel (k >= 7): variants of libdqhip.so with its LDS reads, its barriers or
# its global fetches removed (results WRONG, only the time counts) as csrc/build/ablate/libdqhip_dense_<tag>.so.
# usage: tools/ablate_dense.sh (here); on the GPU box: DQHIP_LIBRARY=.../libdqhip_dense_nolds.so python tools/bench_dense.py
set -euo pipefail
cd "$(dirname "$0")/.."
csrc=deepquantum_amd/csrc
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value"
mkdir -p $csrc/build/ablate
build() {
  local tag=$1; shift
  $HIPCC $FLAGS "$@" -c $csrc/dq_dense.hip -o $csrc/build/ablate/dq_dense_$tag.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC $csrc/build/dq_capi.o $csrc/build/dq_gate.o $csrc/build/ablate/dq_dense_$tag.o $csrc/build/dq_pass.o \
     $csrc/build/dq_wave.o $csrc/build/dq_reduce.o $csrc/build/dq_dist.o $csrc/build/dq_plan.o -o $csrc/build/ablate/libdqhip_dense_$tag.so
  echo "built libdqhip_dense_$tag.so"
}
build nolds -DDQ_DENSE_ABL_NOLDS &
build nobar -DDQ_DENSE_ABL_NOBAR &
build nofetch -DDQ_DENSE_ABL_NOFETCH &
build nolds_nobar -DDQ_DENSE_ABL_NOLDS -DDQ_DENSE_ABL_NOBAR &
build kc32 -DDQ_DENSE_KC_F32=32 &
wait
