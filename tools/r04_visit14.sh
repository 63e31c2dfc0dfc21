#!/bin/bash
# round-4 visit 14: aggregation with two nodes per wave (tuning key 7 = 9 / 10) against the default, standalone and inside the forward
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v14; rm -rf $O; mkdir -p $O
timeout 300 python tools/agg_time.py 128 variants 2>&1 | grep -v amdgpu.ids | grep "round 2" | tee $O/agg_variants.txt
timeout 300 python tools/forward_ab.py 0,9,11,12,13 7 2>&1 | grep -v amdgpu.ids | tee $O/forward_ab.txt
