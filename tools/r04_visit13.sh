#!/bin/bash
# round-4 visit 13: mode 4 of the fp16x3 H = 256 kernel with the plane conversion in the epilogue waves
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v13; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_edge_tile_f16.py tests/test_hip_parity.py -m gpu -x -q -k "f16 or linear or projection" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log
timeout 300 python tools/linear_time.py 256 2>&1 | grep -v amdgpu.ids | tail -4
timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras > $O/bench_c4shard.json 2> $O/bench_c4shard.err; echo "c4shard rc=$?"
python - $O/bench_c4shard.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(round(d["ms_per_step"],3),"ms gate", round(d["roofline"]["avg_launch_ms"],4), [(k["kernel"][:10], round(k["avg_launch_ms"],4)) for k in d.get("kernels",[])[:4]])
PY
