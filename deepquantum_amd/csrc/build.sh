#!/bin/bash
# Build libdqhip.so for gfx950 (cross-compiles without a GPU).  Output lands next to the package so it
# travels to the GPU box with the repo snapshot.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libdqhip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
srcs=(dq_capi.hip dq_gate.hip dq_dense.hip dq_pass.hip dq_wave.hip dq_reduce.hip dq_dist.hip dq_plan.hip)
objs=()
mkdir -p "${here}/build"
pids=()
for s in "${srcs[@]}"; do
  o="${here}/build/${s%.hip}.o"
  objs+=("$o")
  if [[ ! -f "$o" || "${here}/$s" -nt "$o" || "${here}/dq_common.hpp" -nt "$o" || "${here}/../../include/dq_hip.h" -nt "$o" || ( "$s" == dq_wave.hip && ( "${here}/dq_wave_asm.inc" -nt "$o" || "${here}/dq_wave_asm64.inc" -nt "$o" ) ) ]]; then
    if [[ "$s" == dq_wave.hip ]]; then
      "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -mllvm -simplifycfg-sink-common=false -mllvm -structurizecfg-skip-uniform-regions -Rpass-analysis=kernel-resource-usage ${DQ_HIPCC_EXTRA:-} -c "${here}/$s" -o "$o" 2> "${here}/build/dq_wave.usage" &
    else
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Wno-unused-value -mllvm -simplifycfg-sink-common=false -mllvm -structurizecfg-skip-uniform-regions ${DQ_HIPCC_EXTRA:-} -c "${here}/$s" -o "$o" &
    fi
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do
  if [[ -n "$p" ]] && ! wait "$p"; then
    grep -B2 -A3 "error" "${here}/build/dq_wave.usage" 2>/dev/null | head -40 >&2
    echo "build failed" >&2
    exit 1
  fi
done
# The wave-tile kernels are built for three waves per SIMD: 168 VGPRs, nothing spilled.  One SGPR too many held across
# the assembly spills into a 169th VGPR and silently costs a third of the occupancy (8 % of the headline, measured).
if [[ -f "${here}/build/dq_wave.usage" ]]; then
  if grep -A8 "wave_pass_kernel" "${here}/build/dq_wave.usage" | grep -E "VGPRs: (169|1[7-9][0-9]|[2-9][0-9][0-9])|Spill: [1-9]" > /dev/null; then
    echo "dq_wave.hip: a wave_pass_kernel instantiation exceeds 168 VGPRs or spills (see build/dq_wave.usage)" >&2
    grep -E "Function Name|VGPRs:|Spill" "${here}/build/dq_wave.usage" | sed 's/.*remark: //' >&2
    exit 1
  fi
fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out"
