#!/bin/bash
# soak + race hunt on the final build (the kernels whose request order / MFMA schedule changed late in round 4)
mkdir -p gpurun_out/v39
O=gpurun_out/v39
timeout 900 python tools/gate_soak.py 3 1500 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt | tail -2
timeout 900 python tools/gate_soak.py 41 1500 2>&1 | grep -v amdgpu.ids | tee -a $O/soak.txt | tail -2
timeout 900 python tools/f16_repeat.py 2500000 200 2>&1 | grep -v amdgpu.ids | tee -a $O/soak.txt
