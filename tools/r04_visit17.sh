#!/bin/bash
# round-4 visit 17: reference-order kernels at K / H = 256 (matrix-core form) + the synthetic high-gain checkpoint
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v17; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_reference_order.py -m gpu -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -12 $O/pytest.log | cut -c1-300
grep high_gain gpurun_out/parity_margins.jsonl parity_margins.jsonl 2>/dev/null | tail -2
