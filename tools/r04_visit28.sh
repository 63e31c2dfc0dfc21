#!/bin/bash
# round-4 visit 28: the two-rank plumbing line of configs[4]'s shard with bf16 activation storage on the partition
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v28; rm -rf $O; mkdir -p $O
for st in fp32 bf16; do
timeout 900 python bench.py --gpus 2 --one-gpu-gloo --mode train --workload c4shard --steps 3 --warmup 1 --storage $st > $O/n2_train_c4shard_$st.json 2> $O/n2_$st.err; echo "$st rc=$?"
python - $O/n2_train_c4shard_$st.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(round(d["ms_per_step"],1),"ms loss",d["loss"],"storage",d.get("activation_storage"),"in step",d["ranks_in_step"],"rank0 peak GB",round(d["rank0"]["peak_memory_GB"],1))
except Exception as ex: print("FAILED",ex)
PY
tail -2 $O/n2_$st.err | grep -v amdgpu
done
