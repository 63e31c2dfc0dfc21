#!/bin/bash
# kernel table of the H = 256 training step at the 2.5M-edge shard of configs[3] (rocprofv3 --kernel-trace --stats)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/t256; rm -rf $O; mkdir -p $O
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python bench.py --workload c4shard --mode train --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-extras > $O/bench.json 2> $O/err
python tools/rocpd_summary.py "$(find $O/prof -name '*.db' | head -1)" > $O/${1:-r03}_c4shard_train.kernel_stats.md 2>&1; rm -rf $O/prof
head -40 $O/${1:-r03}_c4shard_train.kernel_stats.md | cut -c1-160
