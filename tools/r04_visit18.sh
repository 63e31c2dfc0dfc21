#!/bin/bash
# round-4 visit 18: fp16x3 streaming scorer (H = 64 / 128 / 256), 200k-edge oracle-autograd test; A/B of c2 / c4shard / ecoli / parity64
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v18; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_reference_order.py tests/test_hip_training.py -m gpu -q -k "edge_score or goldens or ecoli or golden or 200k or layer_and or properties" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -8 $O/pytest.log | cut -c1-300
for w in c2 c4shard ecoli parity64; do for t in "" "10=1"; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-extras ${t:+--tuning $t} > $O/b.json 2> $O/b.err
  python - $O/b.json "$w [$t]" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], round(d["ms_per_step"],4),"ms", [(k["kernel"][:12], round(k["avg_launch_ms"],4)) for k in d.get("kernels",[])[:4]])
except Exception as ex: print(sys.argv[2],"FAILED",ex)
PY
done; done
