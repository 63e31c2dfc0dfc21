#!/bin/bash
# K = 128 projection: stores from the compute waves (tuning 11 = 0, new default) against the x-tile route (11 = 1)
mkdir -p gpurun_out/v30
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "linear or goldens or oracle_mid or world1" > gpurun_out/v30/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v30/pytest.log
tail -3 gpurun_out/v30/pytest.log
for n in 100000 100003 1000000; do timeout 200 python tools/linear_time.py 128 $n 0,1 11 2>&1 | grep -v amdgpu.ids | sed "s/^/n=$n /"; done > gpurun_out/v30/linear_ab.txt
cat gpurun_out/v30/linear_ab.txt
for rnd in 0 1; do
  for t in 0 1; do
    timeout 300 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --tuning 11=$t 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $rnd tuning 11=$t ms_per_step', round(d['ms_per_step'],4))"
  done
done > gpurun_out/v30/forward_ab.txt 2>&1
cat gpurun_out/v30/forward_ab.txt
