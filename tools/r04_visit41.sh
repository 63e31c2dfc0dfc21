#!/bin/bash
mkdir -p gpurun_out/v41
timeout 900 python -m pytest tests/test_hip_training.py tests/test_hip_parity.py -x -q -m gpu -k "widths" > gpurun_out/v41/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v41/pytest.log
tail -15 gpurun_out/v41/pytest.log
