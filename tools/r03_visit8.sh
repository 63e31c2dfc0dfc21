#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v8; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_hip_partition.py -m gpu -x -q -k "aggregate or hub or captured or hipgraph or golden or reproducible or replay or partition" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --workload ecoli --no-cpu-baseline > $O/bench_ecoli.json 2> $O/bench_ecoli.err; echo "ecoli rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python - <<'PY'
import json
for f in ('ecoli','c2'):
    d=json.load(open(f'gpurun_out/v8/bench_{f}.json'))
    print(f, d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])
PY
