#!/bin/bash
# round-4 visit 9: fp16x3 in the H = 128 plane form (modes 0, 1, 4): tests, c2 forward / training A/B against bf16x6 on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v9; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_reference_order.py -m gpu -x -q > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -8 $O/pytest_parity.log
for t in "" "10=1" "" "10=1"; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras ${t:+--tuning $t} > $O/b.json 2> $O/b.err; echo "c2 [$t] rc=$?"
  python - $O/b.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("  ",round(d["ms_per_step"],4),"ms", round(d["value"]/1e6,1),"M edges/s gate", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],3), [ (k["kernel"][:12], round(k["avg_launch_ms"],4)) for k in d.get("kernels",[])[:5]])
except Exception as ex:
    print("FAILED",ex)
PY
done
timeout 300 python tools/linear_time.py 128 2>&1 | grep -v amdgpu.ids | tail -6
for t in "" "10=1"; do
  timeout 400 python bench.py --mode train --no-cpu-baseline ${t:+--tuning $t} > $O/t.json 2> $O/t.err; echo "c2 train [$t] rc=$?"
  python - $O/t.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("  ",round(d["ms_per_step"],3),"ms eager", d.get("eager_ms_per_step"), "loss", d.get("loss"))
except Exception as ex:
    print("FAILED",ex)
PY
done
