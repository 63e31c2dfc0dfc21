#!/bin/bash
# H = 256 edge-tile kernel: A/B of two builds (build/ab/prev.so, new.so)
mkdir -p gpurun_out/v36
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_edge_tile_f16.py -x -q -m gpu -k "256 or f16 or linear or goldens or folded" > gpurun_out/v36/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v36/pytest.log
tail -3 gpurun_out/v36/pytest.log
tools/ab_two_builds.sh 2 bash -c 'python tools/gate_time.py --hidden 256 --edges 2500000 2>&1 | grep "variant 0"; python tools/linear_time.py 256 250000 0 2>&1 | grep "round 2"; python bench.py --workload c4shard --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c4shard forward ms_per_step\", round(d[\"ms_per_step\"],3), \"gate\", round(d[\"roofline\"][\"avg_launch_ms\"],4))"' > gpurun_out/v36/ab.txt 2>&1
cat gpurun_out/v36/ab.txt
python tools/gate_phase_profile.py --hidden 256 --edges 2500000 2>&1 | grep -v amdgpu.ids > gpurun_out/v36/gate256_phases.txt; cat gpurun_out/v36/gate256_phases.txt
