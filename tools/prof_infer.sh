#!/bin/bash
# rocprofv3 kernel trace of the default inference workload -> gpurun_out/<tag>/kernel_stats.md   usage: tools/prof_infer.sh <tag> [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-extras "$@" > $O/bench.json 2> $O/err.log
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" > $O/kernel_stats.md 2>&1
find $O/prof -name "*.db" -delete
head -16 $O/kernel_stats.md
