#!/bin/bash
# round-4 visit 23: soak of the counter-synchronised kernels incl. the fp16x3 H = 256 kernel (modes 0, 1, 4, 5), 2 seeds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v23; rm -rf $O; mkdir -p $O
timeout 900 python tools/gate_soak.py 3 1500 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt | tail -4
timeout 900 python tools/gate_soak.py 17 1500 2>&1 | grep -v amdgpu.ids | tee -a $O/soak.txt | tail -4
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "folded_encoder or soak" 2>&1 | tail -3
