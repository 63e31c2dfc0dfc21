#!/bin/bash
# round-4 visit 6: where the fp16x3 H = 256 gate's time goes - compile-time ablations (wrong results, timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v6; rm -rf $O; mkdir -p $O
timeout 600 python tools/gate_time.py --hidden 256 --edges 2500000 --variants 0 --ablations 0,101,102,103,104,107,108,116,124,131 --reps 20 2>&1 | grep -v amdgpu.ids | tee $O/gate_f16_ablations.txt
