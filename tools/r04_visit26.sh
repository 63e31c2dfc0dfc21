#!/bin/bash
# round-4 visit 26: aggregation U = 2 at H = 256 (new default) - tests; H = 128: whole forward for U = 4 (default) / 2 / 1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -x -q -k "aggregate or h256 or configs3 or goldens" 2>&1 | tail -3
timeout 300 python tools/forward_ab.py 0,5,3 7 2>&1 | grep -v amdgpu.ids
timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras > /tmp/b.json 2> /tmp/b.err
python - /tmp/b.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("c4shard", round(d["ms_per_step"],3),"ms", [(k["kernel"][:10], round(k["avg_launch_ms"],4)) for k in d.get("kernels",[])[:1]])
PY
