#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v3; mkdir -p $O
timeout 900 python -m pytest tests/test_reference_order.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python bench.py --workload ecoli --no-cpu-baseline > $O/bench_ecoli.json 2> $O/bench_ecoli.err; echo "ecoli rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/v3/bench_ecoli.json'))
print(d['value'], d['ms_per_step']); print(json.dumps(d.get('reference_order')))
PY
