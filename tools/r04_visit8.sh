#!/bin/bash
# round-4 visit 8: the whole GPU suite on the fp16x3 build + c4shard forward / training lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v8; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1; echo "full suite rc=$?"; tail -8 $O/pytest_full.log
timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras > $O/bench_c4shard.json 2> $O/bench_c4shard.err; echo "c4shard rc=$?"
timeout 600 python bench.py --workload c4shard --mode train --no-cpu-baseline > $O/bench_train_c4shard.json 2> $O/bench_train_c4shard.err; echo "c4shard train rc=$?"
for f in $O/bench_c4shard.json $O/bench_train_c4shard.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],3),"ms", round(d["value"]/1e6,1),"M edges/s", d.get("eager_ms_per_step"), d.get("loss"))
    for k in d.get("kernels",[])[:8]: print("   ",k["kernel"][:40], round(k["avg_launch_ms"],4))
except Exception as ex:
    print("FAILED",ex)
PY
done
