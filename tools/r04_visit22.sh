#!/bin/bash
# round-4 visit 22: layer 0 at H = 256 with the edge encoder folded into the fp16x3 edge-tile kernel (mode 5)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v22; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "folded_encoder or goldens or h256 or layer_and" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.log | cut -c1-250
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python bench.py --workload c4shard --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-extras > $O/prof.json 2> $O/prof.err
python tools/rocpd_summary.py "$(find $O/prof -name '*.db' | head -1)" > $O/c4shard_infer.kernel_stats.md 2>&1; rm -rf $O/prof
head -12 $O/c4shard_infer.kernel_stats.md | cut -c1-150
timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err
python - $O/b.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(round(d["ms_per_step"],3),"ms gate", round(d["roofline"]["avg_launch_ms"],4), [(k["kernel"][:10], round(k["avg_launch_ms"],4), k["launches"]) for k in d.get("kernels",[])[:4]])
PY
