#!/bin/bash
# round-4 visit 7: fp16x3 gate, plane conversion in the epilogue waves: tests, ablations, c4shard forward
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v7; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_edge_tile_f16.py -m gpu -x -q > $O/pytest_f16.log 2>&1; echo "f16 tests rc=$?"; tail -5 $O/pytest_f16.log
timeout 600 python tools/gate_time.py --hidden 256 --edges 2500000 --variants 0 --ablations 0,101,102,104,107,108,116,124,131 --reps 20 2>&1 | grep -v amdgpu.ids | grep "round 1" | tee $O/gate_f16_ablations.txt
timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras > $O/bench_c4shard.json 2> $O/bench_c4shard.err; echo "c4shard rc=$?"
python - $O/bench_c4shard.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(round(d["ms_per_step"],3),"ms", round(d["value"]/1e6,1),"M edges/s", "gate", round(d["roofline"]["avg_launch_ms"],4))
    for k in d.get("kernels",[])[:8]: print("   ",k["kernel"][:40], round(k["avg_launch_ms"],4))
except Exception as ex:
    print("FAILED",ex)
PY
timeout 300 python tools/gate_phase_profile.py --hidden 256 --edges 2500000 > $O/gate256_phases.txt 2>&1; tail -9 $O/gate256_phases.txt
