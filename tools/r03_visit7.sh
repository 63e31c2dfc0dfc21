#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v7; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_training.py tests/test_hip_partition.py -m gpu -x -q -k "linear or 256 or wgrad" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python bench.py --workload c4shard --no-cpu-baseline --no-extras > $O/bench_c4shard.json 2> $O/bench_c4shard.err; echo "c4shard rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/v7/bench_c4shard.json'))
print('c4shard', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
for k in d.get('kernels', []): print(k['kernel'][:40], k['avg_launch_ms'])
PY
timeout 300 python bench.py --workload c4shard --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_c4shard.json 2> $O/bench_train_c4shard.err; echo "train c4shard rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/v7/bench_train_c4shard.json'))
print('train c4shard', d['value'], d['ms_per_step'], d['hbm_roofline_frac_3xBfwd'])
PY
