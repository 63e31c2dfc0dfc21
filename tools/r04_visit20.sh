#!/bin/bash
# round-4 visit 20: mode 4 of the fp16x3 H = 256 kernel with DMA + conversion in the epilogue waves
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v20; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_edge_tile_f16.py tests/test_hip_parity.py -m gpu -x -q -k "f16 or linear or projection or h256 or layer_and" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log
timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err
python - $O/b.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(round(d["ms_per_step"],3),"ms gate", round(d["roofline"]["avg_launch_ms"],4), [(k["kernel"][:10], round(k["avg_launch_ms"],4)) for k in d.get("kernels",[])[:4]])
PY
timeout 400 python bench.py --workload c4shard --mode train --no-cpu-baseline > $O/t.json 2> $O/t.err
python - $O/t.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("train", round(d["ms_per_step"],2), d.get("loss"))
PY
