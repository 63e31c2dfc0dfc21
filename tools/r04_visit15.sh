#!/bin/bash
# round-4 visit 15: two-pass training forward of the gate (H = 128): kernel test, training tests, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v15; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_training.py tests/test_hip_partition.py tests/test_node_order.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.log | cut -c1-250
timeout 300 python tools/train_two_pass_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/two_pass_ab.txt
timeout 300 python tools/train_two_pass_ab.py bf16 2>&1 | grep -v amdgpu.ids | tee $O/two_pass_ab_bf16.txt
