#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v6; mkdir -p $O
timeout 300 python tools/gate_phase_profile.py --hidden 256 --edges 2500000 2>&1 | grep -v amdgpu.ids | tee $O/phase256.txt
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py -m gpu -x -q -k "edge_gate or streaming or fused or gate_raw or residual" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python bench.py --workload c4shard --no-cpu-baseline --no-extras > $O/bench_c4shard.json 2> $O/bench_c4shard.err; echo "c4shard rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/v6/bench_c4shard.json'))
print('c4shard', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
PY
