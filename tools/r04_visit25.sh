#!/bin/bash
# round-4 visit 25: aggregation items in flight at H = 256 inside the whole forward (tuning key 7: 0 = U 8, 5 = U 2, 3 = U 1)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for t in "" "7=5" "7=3" "" "7=5" "7=3"; do
timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras ${t:+--tuning $t} > /tmp/b.json 2> /tmp/b.err
python - /tmp/b.json "c4shard [$t]" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], round(d["ms_per_step"],3),"ms", [(k["kernel"][:10], round(k["avg_launch_ms"],4)) for k in d.get("kernels",[])[:1]])
except Exception as ex: print(sys.argv[2],"FAILED",ex)
PY
done
