#!/bin/bash
# round-4 visit 4 (the session that ran visits 2-3 was lost with its outputs): banded edit distance tests + timing, node-order tests,
# forward on permuted ids with / without the locality order, the c5shard training test, RCCL tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_node_order.py tests/test_overlap_similarity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -25 $O/pytest.log
timeout 600 python tools/overlap_time.py --full-too 2>&1 | grep -v amdgpu.ids | tee $O/overlap_time.txt
timeout 300 python tools/overlap_time.py --rate 0.01 2>&1 | grep -v amdgpu.ids | tee -a $O/overlap_time.txt
timeout 300 python tools/overlap_time.py --rate 0.001 2>&1 | grep -v amdgpu.ids | tee -a $O/overlap_time.txt
for k in banded permuted uniform; do
  timeout 300 python bench.py --kind $k --no-cpu-baseline --no-extras > $O/bench_c2_$k.json 2> $O/bench_c2_$k.err; echo "c2 $k rc=$?"
done
timeout 300 python bench.py --kind permuted --node-order locality --no-cpu-baseline --no-extras > $O/bench_c2_permuted_locality.json 2> $O/bench_c2_permuted_locality.err; echo "rc=$?"
timeout 400 python bench.py --workload 10m --kind permuted --node-order locality --no-cpu-baseline --no-extras > $O/bench_10m_permuted_locality.json 2> $O/bench_10m_permuted_locality.err; echo "rc=$?"
timeout 400 python bench.py --workload 10m --kind permuted --no-cpu-baseline --no-extras > $O/bench_10m_permuted.json 2> $O/bench_10m_permuted.err; echo "rc=$?"
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,1), "M edges/s", json.dumps(d.get("node_order")))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
done
tail -3 $O/*.err | grep -v amdgpu | tail -20
timeout 400 python bench.py --gpus 2 --one-gpu-gloo --workload c2 --kind permuted --node-order locality --steps 5 --warmup 2 > $O/n2_permuted_locality.json 2> $O/n2_permuted_locality.err; echo "n2 rc=$?"; tail -c 600 $O/n2_permuted_locality.json
timeout 900 python -m pytest tests/test_hip_partition.py -m gpu -x -q -k "rccl" > $O/pytest_rccl.log 2>&1; echo "rccl rc=$?"; tail -5 $O/pytest_rccl.log
timeout 900 python -m pytest tests/test_hip_training.py -m gpu -x -q -k "configs4" > $O/pytest_c5.log 2>&1; echo "c5shard test rc=$?"; tail -5 $O/pytest_c5.log
