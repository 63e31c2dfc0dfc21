#!/bin/bash
# round-4 visit 10: the whole GPU suite on the fp16x3 build (H = 128 and 256)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v10; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; echo "full suite rc=$?"; tail -30 $O/pytest_full.log | cut -c1-300
