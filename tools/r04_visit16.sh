#!/bin/bash
# round-4 visit 16: bf16 activation storage at H = 256 (fp16x3 raw gate, plane-form mode 3, 256 x 256 wgrad): tests + c4shard / c5shard training lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v16; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_training.py -m gpu -x -q -k "bf16" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -8 $O/pytest.log | cut -c1-250
for st in fp32 bf16; do
  timeout 600 python bench.py --workload c4shard --mode train --storage $st --no-cpu-baseline > $O/train_c4shard_$st.json 2> $O/train_c4shard_$st.err; echo "c4shard $st rc=$?"
  python - $O/train_c4shard_$st.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); m=d["memory"]["eager_step"]
    print("  ",round(d["ms_per_step"],2),"ms eager",round(d["eager_ms_per_step"],2),"loss",d["loss"],"alloc/res GB",round(m["peak_memory_GB"],1),round(m["peak_reserved_GB"],1))
except Exception as ex: print("FAILED",ex)
PY
  tail -2 $O/train_c4shard_$st.err | grep -v amdgpu
done
timeout 900 python bench.py --workload c5shard --mode train --storage bf16 --no-cpu-baseline > $O/train_c5shard_bf16.json 2> $O/train_c5shard_bf16.err; echo "c5shard bf16 rc=$?"
python - $O/train_c5shard_bf16.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); m=d["memory"]["eager_step"]
    print("  ",round(d["ms_per_step"],2),"ms eager",round(d["eager_ms_per_step"],2),"loss",d["loss"],"alloc/res GB",round(m["peak_memory_GB"],1),round(m["peak_reserved_GB"],1))
except Exception as ex: print("FAILED",ex)
PY
