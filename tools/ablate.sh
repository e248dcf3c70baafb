#!/bin/bash
# Timing experiments on the fused pass: builds variants of libdqhip.so with parts of the kernel removed (results are
# WRONG, only the time is of interest) as deepquantum_amd/csrc/build/ablate/libdqhip_<tag>.so.
# usage: tools/ablate.sh            (here, no GPU needed)
#        DQHIP_LIBRARY=$PWD/deepquantum_amd/csrc/build/ablate/libdqhip_nogates.so python bench.py ...      (on the GPU box)
set -euo pipefail
cd "$(dirname "$0")/.."
csrc=deepquantum_amd/csrc
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -mllvm -simplifycfg-sink-common=false -mllvm -structurizecfg-skip-uniform-regions"
mkdir -p $csrc/build/ablate
DQ_ABLATE_GATES=1 DQ_ASM_OUT=$PWD/$csrc/build/ablate/nogates_asm.inc python tools/gen_fused_asm.py > /dev/null
build() {   # tag, extra flags
  local tag=$1; shift
  $HIPCC $FLAGS "$@" -c $csrc/dq_fused.hip -o $csrc/build/ablate/dq_fused_$tag.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC $csrc/build/dq_capi.o $csrc/build/dq_gate.o $csrc/build/dq_dense.o $csrc/build/ablate/dq_fused_$tag.o \
     $csrc/build/dq_reduce.o $csrc/build/dq_dist.o -o $csrc/build/ablate/libdqhip_$tag.so
  echo "built $csrc/build/ablate/libdqhip_$tag.so"
}
build nogates "-DDQ_ASM_INC=\"$PWD/$csrc/build/ablate/nogates_asm.inc\"" &
build nolds -DDQ_ABLATE_LDS &
build nogates_nolds "-DDQ_ASM_INC=\"$PWD/$csrc/build/ablate/nogates_asm.inc\"" -DDQ_ABLATE_LDS &
build nobar -DDQ_ABLATE_BARRIER &
wait
