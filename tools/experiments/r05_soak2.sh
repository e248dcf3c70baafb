#!/bin/bash
# Soak of the round's last changes (ABI 24: long sweep passes, one sum per rotation, X-shaped channel bodies, forward mode
# under torch.func): new seeds for every mode of tools/soak.py.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/soak2 && out=gpurun_out/soak2/soak.txt
: > $out
timeout 900 python tools/soak.py 7000 40 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
timeout 600 python tools/soak.py 7000 60 small 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
timeout 600 python tools/soak.py 7000 40 hvp 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
timeout 900 python tools/soak.py 7000 36 func 2>&1 | grep -v amdgpu.ids | tail -1 >> $out
cat $out
