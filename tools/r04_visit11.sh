#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v11; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -q -k "linear or bf16_storage or training_step" > $O/pytest.log 2>&1; echo "rc=$?"; tail -8 $O/pytest.log | cut -c1-300
