#!/bin/bash
mkdir -p gpurun_out/v33
for i in 1 2; do timeout 200 python tools/gate_phase_profile.py --variant 7 --linear --edges 100000 2>&1 | grep -v amdgpu.ids; done > gpurun_out/v33/linear_phases.txt
timeout 200 python tools/gate_phase_profile.py --variant 7 --linear --edges 1000000 2>&1 | grep -v amdgpu.ids >> gpurun_out/v33/linear_phases.txt
cat gpurun_out/v33/linear_phases.txt
