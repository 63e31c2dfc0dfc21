#!/bin/bash
# round-4 visit 19: B2h[dst] once per run of equal destinations in the fp16x3 H = 256 gate: tests + A/B (ablation 132 = fetched for every piece)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v19; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_edge_tile_f16.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/gate_time.py --hidden 256 --edges 2500000 --variants 0 --ablations 0,132 --reps 30 2>&1 | grep -v amdgpu.ids | tee $O/gate_fresh_ab.txt
for t in "" "1=132"; do
timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras ${t:+--tuning $t} > $O/b.json 2> $O/b.err
python - $O/b.json "c4shard [$t]" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], round(d["ms_per_step"],3),"ms gate", round(d["roofline"]["avg_launch_ms"],4))
except Exception as ex: print(sys.argv[2],"FAILED",ex)
PY
done
