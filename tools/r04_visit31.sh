#!/bin/bash
# EXPERIMENT: projection output as five contiguous [N,128] matrices (key 4 = 16) against [N,640] rows
mkdir -p gpurun_out/v31
for n in 100000 1000000; do timeout 200 python tools/linear_time.py 128 $n 0,16 4 2>&1 | grep -v amdgpu.ids | sed "s/^/n=$n /"; done > gpurun_out/v31/linear_blockmajor.txt
cat gpurun_out/v31/linear_blockmajor.txt
