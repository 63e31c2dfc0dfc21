#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/ptrain
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ptrain -o r01t -- python bench.py --mode train --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timers > gpurun_out/ptrain/bench.json 2> gpurun_out/ptrain/err.log
echo "rc=$?"; tail -c 300 gpurun_out/ptrain/bench.json
find gpurun_out/ptrain -name "*.db" | head
DB=$(find gpurun_out/ptrain -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" > gpurun_out/ptrain/kernel_stats.md 2>&1
head -40 gpurun_out/ptrain/kernel_stats.md
find gpurun_out/ptrain -name "*.db" -delete
timeout 300 python bench.py --mode train --hipgraph --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ptrain/bench_hipgraph.json 2>> gpurun_out/ptrain/err.log; tail -c 400 gpurun_out/ptrain/bench_hipgraph.json
