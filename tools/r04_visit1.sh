#!/bin/bash
# round-4 visit 1: RCCL single-rank tests, configs[4]-shard training memory (stored / recomputed xe), H = 256 kernel tables with per-mode rows
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_partition.py -x -q -k "rccl" > $O/pytest_rccl.log 2>&1; echo "rccl tests rc=$?"; tail -15 $O/pytest_rccl.log
for w in c4shard c5shard; do
  for r in "" "--recompute-gate"; do
    tag=${w}${r:+_recompute}
    timeout 900 python bench.py --mode train --workload $w --no-cpu-baseline $r > $O/train_$tag.json 2> $O/train_$tag.err; echo "train $tag rc=$?"
    python - "$O/train_$tag.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    m=d["memory"]
    print(round(d["ms_per_step"],1),"ms eager",round(d["eager_ms_per_step"],1),"| eager alloc/res GB",round(m["eager_step"]["peak_memory_GB"],1),round(m["eager_step"]["peak_reserved_GB"],1),"B/edge",round(m["eager_step"]["bytes_per_local_edge_reserved"]),"| graph alloc/res",round(m["hipgraph_step"]["peak_memory_GB"],1),round(m["hipgraph_step"]["peak_reserved_GB"],1),"loss",d["loss"])
except Exception as ex:
    print("FAILED",ex)
PY
    tail -3 $O/train_$tag.err
  done
done
timeout 900 python bench.py --gpus 2 --one-gpu-gloo --mode train --workload c5shard --steps 3 --warmup 1 --recompute-gate > $O/n2_train_c5shard.json 2> $O/n2_train_c5shard.err; echo "n2 rc=$?"; tail -c 700 $O/n2_train_c5shard.json
# kernel tables at H = 256 with the fixed summary (modes as separate rows)
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o r -- python bench.py --workload c4shard --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-extras > /dev/null 2> $O/prof_c4.err
python tools/rocpd_summary.py "$(find $O/prof_c4 -name '*.db' | head -1)" > $O/r04_c4shard_infer.kernel_stats.md 2>&1; rm -rf $O/prof_c4
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_t -o r -- python bench.py --workload c4shard --mode train --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-extras > /dev/null 2> $O/prof_t.err
python tools/rocpd_summary.py "$(find $O/prof_t -name '*.db' | head -1)" > $O/r04_c4shard_train.kernel_stats.md 2>&1; rm -rf $O/prof_t
head -30 $O/r04_c4shard_infer.kernel_stats.md | cut -c1-200
head -45 $O/r04_c4shard_train.kernel_stats.md | cut -c1-200
# the full GPU suite on this build
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_full.log 2>&1; echo "full suite rc=$?"; tail -5 $O/pytest_full.log
