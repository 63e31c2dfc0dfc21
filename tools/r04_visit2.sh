#!/bin/bash
# round-4 visit 2: banded edit distance (tests + timing), the c5shard training test, RCCL tests with the sliced plan
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_overlap_similarity.py -m gpu -x -q > $O/pytest_overlap.log 2>&1; echo "overlap tests rc=$?"; tail -15 $O/pytest_overlap.log
timeout 600 python tools/overlap_time.py --full-too 2>&1 | grep -v amdgpu.ids | tee $O/overlap_time.txt
timeout 300 python tools/overlap_time.py --rate 0.01 2>&1 | grep -v amdgpu.ids | tee -a $O/overlap_time.txt
timeout 300 python tools/overlap_time.py --rate 0.0 2>&1 | grep -v amdgpu.ids | tee -a $O/overlap_time.txt
timeout 900 python -m pytest tests/test_hip_partition.py -m gpu -x -q -k "rccl" > $O/pytest_rccl.log 2>&1; echo "rccl rc=$?"; tail -5 $O/pytest_rccl.log
timeout 900 python -m pytest tests/test_hip_training.py -m gpu -x -q -k "configs4" > $O/pytest_c5.log 2>&1; echo "c5shard test rc=$?"; tail -5 $O/pytest_c5.log
timeout 600 python bench.py --gpus 2 --one-gpu-gloo --workload c2 --steps 5 --warmup 2 > $O/n2_infer.json 2> $O/n2_infer.err; echo "n2 infer rc=$?"; tail -c 300 $O/n2_infer.json; tail -3 $O/n2_infer.err
