#!/bin/bash
# plane-form MFMA loop with plane fragments three k steps ahead (new) against hipcc's two-MFMA distance (prev)
mkdir -p gpurun_out/v35
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_edge_tile_f16.py -x -q -m gpu -k "linear or edge_gate or goldens or oracle_mid or world1 or folded or soak" > gpurun_out/v35/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v35/pytest.log
tail -3 gpurun_out/v35/pytest.log
tools/ab_two_builds.sh 2 bash -c 'python tools/linear_time.py 128 100000 0 2>&1 | grep "round 2"; python tools/linear_time.py 128 1000000 0 2>&1 | grep "round 2"; python tools/gate_time.py --hidden 128 2>&1 | tail -2; python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c2 forward ms_per_step\", round(d[\"ms_per_step\"],4), \"gate\", round(d[\"roofline\"][\"avg_launch_ms\"],4))"' > gpurun_out/v35/ab.txt 2>&1
cat gpurun_out/v35/ab.txt
python tools/gate_phase_profile.py --variant 7 --linear --edges 100000 2>&1 | grep -v amdgpu.ids > gpurun_out/v35/linear_phases.txt; cat gpurun_out/v35/linear_phases.txt
