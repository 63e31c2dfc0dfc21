#!/bin/bash
# padded widths on the GPU + the bounds of the two unbuilt training fusions
mkdir -p gpurun_out/v29
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "widths or layer_and_predictor or goldens or oracle_mid" > gpurun_out/v29/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v29/pytest.log
timeout 300 python tools/train_fusion_bounds.py 128 > gpurun_out/v29/fusion_bounds.txt 2>&1
tail -5 gpurun_out/v29/pytest.log; cat gpurun_out/v29/fusion_bounds.txt
