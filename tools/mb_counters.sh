#!/bin/bash
# The stand-alone microbenchmark of the wave-tile design (tools/experiments/mb_wavetile.hip: loads, gates, trips, stores
# with the write patterns of real passes, plain and streaming accesses) and the VALU issue-cost table (mb_valu.hip).
cd "$(dirname "$0")/experiments"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result mb_wavetile.hip -o /tmp/mb_wavetile 2>/dev/null && /tmp/mb_wavetile
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result mb_valu.hip -o /tmp/mb_valu 2>/dev/null && /tmp/mb_valu
