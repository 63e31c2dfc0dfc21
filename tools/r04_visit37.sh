#!/bin/bash
# k_edge_gate_enc16 (layer 0 at H = 128): gathers of the next tile ahead of the epilogue, indices two tiles ahead (new) against the closing build (prev)
mkdir -p gpurun_out/v37
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "folded or goldens or oracle_mid or world1 or soak or full_size or reversed" > gpurun_out/v37/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v37/pytest.log
tail -3 gpurun_out/v37/pytest.log
tools/ab_two_builds.sh 2 bash -c 'python tools/enc256_time.py 128 2>&1 | grep "round 1"; python tools/enc256_time.py 128 10000000 2>&1 | grep "round 1"; python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c2 forward ms_per_step\", round(d[\"ms_per_step\"],4))"' > gpurun_out/v37/ab.txt 2>&1
cat gpurun_out/v37/ab.txt
