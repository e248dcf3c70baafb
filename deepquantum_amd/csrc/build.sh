#!/bin/bash
# Build libdqhip.so for gfx950 (cross-compiles without a GPU).  Output lands next to the package so it
# travels to the GPU box with the repo snapshot.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libdqhip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
srcs=(dq_capi.hip dq_gate.hip dq_dense.hip dq_fused.hip dq_wave.hip dq_reduce.hip dq_dist.hip dq_plan.hip)
objs=()
mkdir -p "${here}/build"
pids=()
for s in "${srcs[@]}"; do
  o="${here}/build/${s%.hip}.o"
  objs+=("$o")
  if [[ ! -f "$o" || "${here}/$s" -nt "$o" || "${here}/dq_common.hpp" -nt "$o" || "${here}/../../include/dq_hip.h" -nt "$o" || ( "$s" == dq_fused.hip && "${here}/dq_fused_asm.inc" -nt "$o" ) || ( "$s" == dq_wave.hip && ( "${here}/dq_wave_asm.inc" -nt "$o" || "${here}/dq_wave_asm64.inc" -nt "$o" ) ) ]]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -mllvm -simplifycfg-sink-common=false -mllvm -structurizecfg-skip-uniform-regions ${DQ_HIPCC_EXTRA:-} -c "${here}/$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out"
