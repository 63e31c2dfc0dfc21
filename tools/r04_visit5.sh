#!/bin/bash
# round-4 visit 5: the fp16x3 + LDS-DMA edge-tile kernel at H = 256 (edge_tile_f16.hip): tests, then A/B against the bf16x6 plane form
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v5; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_edge_tile_f16.py -m gpu -x -q > $O/pytest_f16.log 2>&1; echo "f16 tests rc=$?"; tail -15 $O/pytest_f16.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "edge_gate or h256 or linear or layer_and or goldens" > $O/pytest_parity.log 2>&1; echo "parity subset rc=$?"; tail -8 $O/pytest_parity.log
for t in "" "10=1"; do
  tag=${t:+_bf16x6}
  timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras ${t:+--tuning $t} > $O/bench_c4shard$tag.json 2> $O/bench_c4shard$tag.err; echo "c4shard$tag rc=$?"
  python - $O/bench_c4shard$tag.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(round(d["ms_per_step"],3),"ms", round(d["value"]/1e6,1),"M edges/s", json.dumps(d["roofline"])[:400])
    for k in d.get("kernels",[])[:8]: print("   ",json.dumps(k)[:300])
except Exception as ex:
    print("FAILED",ex)
PY
  tail -3 $O/bench_c4shard$tag.err | grep -v amdgpu
done
timeout 300 python tools/gate_phase_profile.py --hidden 256 --edges 2500000 > $O/gate256_phases.txt 2>&1; tail -12 $O/gate256_phases.txt
