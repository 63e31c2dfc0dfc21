#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v5; mkdir -p $O
timeout 300 python tools/gate_phase_profile.py --hidden 256 --edges 2500000 > $O/phase256.txt 2>&1; cat $O/phase256.txt
timeout 900 python -m pytest tests/test_hip_training.py tests/test_hip_partition.py tests/test_hip_parity.py -m gpu -x -q -k "256 or fused or gate_raw or residual or streaming" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python bench.py --workload c4shard --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_c4shard.json 2> $O/bench_train_c4shard.err; echo "train c4shard rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/v5/bench_train_c4shard.json'))
print('train c4shard', d['value'], d['ms_per_step'], d['hbm_roofline_frac_3xBfwd'])
PY
tail -3 $O/bench_train_c4shard.err
timeout 300 python bench.py --workload c4shard --no-cpu-baseline --no-extras > $O/bench_c4shard.json 2> $O/bench_c4shard.err; echo "c4shard rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/v5/bench_c4shard.json'))
print('c4shard', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
PY
