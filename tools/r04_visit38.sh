#!/bin/bash
# aggregation: one memory round trip for everything that depends on the CSR pointers alone (new) against prev
mkdir -p gpurun_out/v38
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "aggregate or goldens or oracle_mid or world1 or full_size or reversed or empty or capture" > gpurun_out/v38/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v38/pytest.log
tail -3 gpurun_out/v38/pytest.log
tools/ab_two_builds.sh 2 bash -c 'python tools/agg_time.py 128 variants 2>&1 | grep "round 2 variant 0"; python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c2 forward ms_per_step\", round(d[\"ms_per_step\"],4))"; python bench.py --workload c4shard --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c4shard forward ms_per_step\", round(d[\"ms_per_step\"],3))"' > gpurun_out/v38/ab.txt 2>&1
cat gpurun_out/v38/ab.txt
