#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v4; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_reference_order.py -m gpu -x -q -k "edge_gate or h256 or streaming or reference or matrix_core or linear" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python bench.py --workload ecoli --no-cpu-baseline > $O/bench_ecoli.json 2> $O/bench_ecoli.err; echo "ecoli rc=$?"
timeout 300 python bench.py --workload c4shard --no-cpu-baseline --no-extras > $O/bench_c4shard.json 2> $O/bench_c4shard.err; echo "c4shard rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/v4/bench_ecoli.json'))
print('ecoli', d['value'], d['ms_per_step']); print(json.dumps(d.get('reference_order')))
d=json.load(open('gpurun_out/v4/bench_c4shard.json'))
print('c4shard', d['value'], d['ms_per_step']); print(json.dumps(d.get('roofline')))
for k in d.get('kernels', []): print(k['kernel'][:40], k['avg_launch_ms'])
PY
tail -3 $O/bench_c4shard.err
