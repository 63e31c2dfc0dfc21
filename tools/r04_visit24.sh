#!/bin/bash
# round-4 visit 24: race hunt for the fp16x3 kernel at full size (bit-identical repeats), extra soak seeds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v24; rm -rf $O; mkdir -p $O
timeout 900 python tools/f16_repeat.py 2500000 300 2>&1 | grep -v amdgpu.ids | tee $O/f16_repeat.txt
timeout 600 python tools/f16_repeat.py 6250000 60 2>&1 | grep -v amdgpu.ids | tee -a $O/f16_repeat.txt
timeout 900 python tools/gate_soak.py 29 2000 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/f16_repeat.txt
