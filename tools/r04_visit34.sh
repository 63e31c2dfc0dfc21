#!/bin/bash
# projection: A rows of the next tile requested ahead of the epilogue (default) against after it (key 4 = 32)
mkdir -p gpurun_out/v34
for n in 100000 1000000; do timeout 200 python tools/linear_time.py 128 $n 0,32 4 2>&1 | grep -v amdgpu.ids | sed "s/^/n=$n /"; done > gpurun_out/v34/linear_fetch_ahead.txt
cat gpurun_out/v34/linear_fetch_ahead.txt
timeout 200 python tools/gate_phase_profile.py --variant 7 --linear --edges 100000 2>&1 | grep -v amdgpu.ids > gpurun_out/v34/linear_phases.txt; cat gpurun_out/v34/linear_phases.txt
for rnd in 0 1; do
  for t in 0 32; do
    timeout 300 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --tuning 4=$t 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $rnd tuning 4=$t ms_per_step', round(d['ms_per_step'],4))"
  done
done > gpurun_out/v34/forward_ab.txt 2>&1
cat gpurun_out/v34/forward_ab.txt
