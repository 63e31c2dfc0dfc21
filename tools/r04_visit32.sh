#!/bin/bash
mkdir -p gpurun_out/v32
hipcc -O3 --offload-arch=gfx950 tools/mem_bandwidth.hip -o /tmp/mem_bandwidth 2>/dev/null && timeout 300 /tmp/mem_bandwidth > gpurun_out/v32/mem_bandwidth.txt 2>&1
tail -14 gpurun_out/v32/mem_bandwidth.txt
