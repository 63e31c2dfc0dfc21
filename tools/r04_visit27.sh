#!/bin/bash
# round-4 visit 27: bf16 activation storage on partitions (two ranks sharing the GPU) + the c5shard two-rank plumbing line with it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v27; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_partition.py -m gpu -x -q -k "bf16 or h256_matches" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.log | cut -c1-250
