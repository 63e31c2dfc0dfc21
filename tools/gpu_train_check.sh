#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_hip_training.py tests/test_hip_partition.py -x -q 2>&1 | tail -15
timeout 300 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/train_now.json 2>gpurun_out/train_now.err; tail -c 700 gpurun_out/train_now.json
