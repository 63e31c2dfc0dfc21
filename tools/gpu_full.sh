#!/bin/bash
# full GPU validation + benches (each leg bounded)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/full/pytest.log
for w in c2 10m parity64 c4shard; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline > gpurun_out/full/bench_$w.json 2>gpurun_out/full/bench_$w.err
  python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
try:
    d=json.loads([l for l in open(f"gpurun_out/full/bench_{w}.json") if l.startswith("{")][-1])
    print(w, round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,1), "M edges/s gate", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],3), "host", round(d["host_enqueue_ms_per_step"],2))
except Exception as ex:
    print(w, "FAILED", ex)
PY
done
timeout 400 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/full/bench_train.json 2>gpurun_out/full/bench_train.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/full/bench_train.json") if l.startswith("{")][-1])
    print("train", round(d["ms_per_step"],2), "ms", round(d["value"]/1e6,2), "M edges/s host", round(d["host_enqueue_ms_per_step"],1))
except Exception as ex:
    print("train FAILED", ex)
PY
